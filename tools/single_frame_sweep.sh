# Frame-by-frame rate of two builds of the library at several frame sizes (run on the GPU box): usage: bash tools/single_frame_sweep.sh "<lib> <lib>"
# (a library is librtx_hip_<name>.so beside the package's own, or "base" for that one); tools/single_frame_probe.py does the measuring.
for s in "480 270" "960 540" "1360 768" "1920 1080" "2560 1440" "3840 2160"; do set -- $s
  for l in ${LIBS:-base}; do
    if [ "$l" = base ]; then L=""; else L="--lib librtx_hip_$l.so"; fi
    echo "size $1 x $2 lib=$l"; timeout -k 10 100 python -u tools/single_frame_probe.py --width $1 --height $2 --frames 200 --modes 00,00,00 $L | tail -n 2 || exit 1
  done
done
