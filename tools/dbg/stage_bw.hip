// micro-benchmark: how fast can one-wave workgroups stage 9 KB tiles into LDS?  (tools only)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define TL_TILE 8192u
#define TL_HALO 1024u
template <int WAVES>
__global__ void __launch_bounds__(64 * WAVES) stage_only(const uint8_t *u, uint64_t ulen, int64_t ntiles, uint32_t *out) {
    __shared__ __attribute__((aligned(16))) uint8_t buf[WAVES][TL_TILE + TL_HALO];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t t = (int64_t)blockIdx.x * WAVES + w;
    if (t >= ntiles) return;
    const uint64_t tb = (uint64_t)t * TL_TILE;
    uint64_t left = ulen - tb; uint32_t avail = left < (TL_TILE + TL_HALO) ? (uint32_t)left : (TL_TILE + TL_HALO);
    uint32_t pad = (avail + 15u) & ~15u;
    for (uint32_t k = (uint32_t)lane * 16u; k < pad; k += 1024u) { uint4 v = *(const uint4 *)(u + tb + k); *(uint4 *)(buf[w] + k) = v; }
    __syncthreads();
    uint32_t x = 0; for (int k = lane * 4; k < 8192; k += 256 * 8) x ^= *(const uint32_t *)(buf[w] + k);
    if (x == 0x12345678u) out[t] = x;
}
__global__ void __launch_bounds__(256) copy_read(const uint4 *u, uint64_t n16, uint32_t *out) {
    uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; uint32_t x = 0;
    for (; i < n16; i += (uint64_t)gridDim.x * 256) { uint4 v = u[i]; x ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (x == 0x12345678u) out[0] = x;
}
int main() {
    const uint64_t ulen = 1072ull << 20; uint8_t *u; uint32_t *out;
    hipMalloc(&u, ulen + 4096); hipMalloc(&out, (ulen / TL_TILE + 16) * 4); hipMemset(u, 1, ulen + 4096);
    const int64_t nt = ulen / TL_TILE;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); float ms;
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(a); hipLaunchKernelGGL(stage_only<1>, dim3(nt), dim3(64), 0, 0, u, ulen, nt, out); hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
        printf("stage_only<1 wave/WG>  %.3f ms  %.2f TB/s\n", ms, ulen * 1.125 / ms / 1e9);
        hipEventRecord(a); hipLaunchKernelGGL(stage_only<4>, dim3((nt + 3) / 4), dim3(256), 0, 0, u, ulen, nt, out); hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
        printf("stage_only<4 waves/WG> %.3f ms  %.2f TB/s\n", ms, ulen * 1.125 / ms / 1e9);
        hipEventRecord(a); hipLaunchKernelGGL(copy_read, dim3(256 * 16), dim3(256), 0, 0, (const uint4 *)u, ulen / 16, out); hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
        printf("grid-stride read       %.3f ms  %.2f TB/s\n", ms, ulen / ms / 1e9);
    }
    return 0;
}
