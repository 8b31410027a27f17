#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void k(uint32_t *out, uint64_t *cyc) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[4096];
    int lane = threadIdx.x;
    for (int i = lane; i < 4096; i += 64) lds[i] = (uint8_t)(i * 7 + 3);
    __syncthreads();
    uint32_t off = lane * 5 + 1;          // odd, mixed alignment
    uint32_t v, w; uint64_t q;
    uint32_t addr = (uint32_t)(uintptr_t)lds + off;
    asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    asm volatile("ds_read_u16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(w) : "v"(addr) : "memory");
    asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(q) : "v"(addr) : "memory");
    out[lane] = v; out[64 + lane] = w; out[128 + lane] = (uint32_t)q; out[192 + lane] = (uint32_t)(q >> 32);
    // unaligned write
    __syncthreads();
    uint32_t waddr = (uint32_t)(uintptr_t)lds + 2048 + lane * 7 + 3;
    uint32_t val = 0xA0B0C0D0u + lane;
    asm volatile("ds_write_b32 %0, %1\n s_waitcnt lgkmcnt(0)" :: "v"(waddr), "v"(val) : "memory");
    __syncthreads();
    uint32_t r = 0; for (int b = 0; b < 4; b++) r |= (uint32_t)lds[2048 + lane * 7 + 3 + b] << (8 * b);
    out[256 + lane] = r;
    // timing: 64 unaligned b32 reads vs aligned
    uint64_t t0 = clock64(); uint32_t acc = 0;
    for (int it = 0; it < 64; it++) { uint32_t a2 = (uint32_t)(uintptr_t)lds + ((off + it * 13) & 2047); uint32_t x; asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(x) : "v"(a2) : "memory"); acc += x; }
    uint64_t t1 = clock64();
    for (int it = 0; it < 64; it++) { uint32_t a2 = (uint32_t)(uintptr_t)lds + (((off + it * 13) & 2047) & ~3u); uint32_t x; asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(x) : "v"(a2) : "memory"); acc += x; }
    uint64_t t2 = clock64();
    if (lane == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; }
    out[320 + lane] = acc;
}
int main() {
    uint32_t *d; uint64_t *c; hipMalloc(&d, 4096); hipMalloc(&c, 64);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, c);
    uint32_t h[384]; uint64_t hc[2]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost); hipMemcpy(hc, c, 16, hipMemcpyDeviceToHost);
    int bad32 = 0, bad16 = 0, bad64 = 0, badw = 0;
    for (int lane = 0; lane < 64; lane++) {
        uint32_t off = lane * 5 + 1; uint64_t e = 0; for (int b = 0; b < 8; b++) e |= (uint64_t)(uint8_t)((off + b) * 7 + 3) << (8 * b);
        if (h[lane] != (uint32_t)e) bad32++;
        if (h[64 + lane] != (uint32_t)(e & 0xffff)) bad16++;
        if (h[128 + lane] != (uint32_t)e || h[192 + lane] != (uint32_t)(e >> 32)) bad64++;
        if (h[256 + lane] != 0xA0B0C0D0u + lane) badw++;
    }
    printf("unaligned LDS: b32 bad=%d u16 bad=%d b64 bad=%d write_b32 bad=%d ; cycles unaligned=%llu aligned=%llu (64 reads)\n", bad32, bad16, bad64, badw, (unsigned long long)hc[0], (unsigned long long)hc[1]);
    printf("lane1: got %08x\n", h[1]);
    return 0;
}
