// calib_fetch -- calibrates rocprofv3's FETCH_SIZE for this library's access pattern.
// Every lane reads one 64-byte record (four 16-byte loads, as a BvhNode fetch does) at a
// pseudo-random 64-byte-aligned offset of a 4 GiB buffer, far larger than the 256 MiB
// Infinity Cache, so (almost) every record comes from HBM. Requested bytes are known:
// records * 64. Run under `rocprofv3 --pmc FETCH_SIZE` and compare.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k_gather64(const uint4* __restrict__ buf, uint64_t n_rec, uint64_t n_lanes, uint4* __restrict__ out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_lanes) return;
    uint64_t z = (i + 0x9E3779B97F4A7C15ull) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    const uint4* p = buf + (z % n_rec) * 4;
    uint4 a = p[0], b = p[1], c = p[2], d = p[3];
    uint4 r = make_uint4(a.x ^ b.x ^ c.x ^ d.x, a.y ^ b.y ^ c.y ^ d.y, a.z ^ b.z ^ c.z ^ d.z, a.w ^ b.w ^ c.w ^ d.w);
    if (r.x == 0x12345678u) out[i & 1023] = r;   // keeps the loads alive
}
__global__ void k_stream(const uint4* __restrict__ buf, uint64_t n_quads, uint4* __restrict__ out) {
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_quads; i += (uint64_t)gridDim.x * blockDim.x) {
        uint4 v = buf[i];
        acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
    }
    if (acc.x == 0x12345678u) out[threadIdx.x] = acc;
}
int main() {
    const uint64_t bytes = 4ull << 30, n_rec = bytes / 64, n_lanes = 16ull << 20;
    uint4 *buf, *out;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 1 << 20) != hipSuccess) return 1;
    hipMemset(buf, 1, bytes);
    hipDeviceSynchronize();
    k_gather64<<<(unsigned)(n_lanes / 256), 256>>>(buf, n_rec, n_lanes, out);
    k_stream<<<8192, 256>>>(buf, bytes / 16, out);
    hipDeviceSynchronize();
    std::printf("k_gather64 requested_bytes %llu\nk_stream requested_bytes %llu\n", (unsigned long long)(n_lanes * 64), (unsigned long long)bytes);
    return 0;
}
