// valu_issue.hip -- micro-benchmark (test tooling): how many wave64 integer VALU instructions per cycle one SIMD of gfx950 issues
// with 1, 2, 4, 8 resident waves, for the instruction kinds the inflate kernels are made of.  Prints cycles per wave-instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int KIND>
__global__ void __launch_bounds__(64) k_issue(uint32_t *out, int iters, uint32_t seed) {
    __shared__ uint32_t lds[2048];
    const int lane = threadIdx.x;
    for (int i = lane; i < 2048; i += 64) lds[i] = (i * 2654435761u) >> 21;
    __syncthreads();
    uint32_t a0 = seed + lane, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    const long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
        if (KIND == 0) {          // 8 independent v_add_u32
#pragma unroll
            for (int u = 0; u < 8; u++) { a0 += a1; a1 += a2; a2 += a3; a3 += a4; a4 += a5; a5 += a6; a6 += a7; a7 += a0; }
        } else if (KIND == 1) {   // one dependent chain of v_add_u32
#pragma unroll
            for (int u = 0; u < 64; u++) a0 += a1;
            asm volatile("" : "+v"(a0));
        } else if (KIND == 2) {   // mix: and, shift, bfe, alignbit, cndmask (independent chains)
#pragma unroll
            for (int u = 0; u < 8; u++) {
                a0 = a0 & a1; a1 = a1 >> (a2 & 31); a2 = __builtin_amdgcn_ubfe(a2, a3 & 31, 5); a3 = __builtin_amdgcn_alignbit(a3, a4, a5 & 31);
                a4 = (a4 > a5) ? a6 : a7; a5 = a5 | a6; a6 = a6 ^ a7; a7 = a7 + a0;
            }
        } else if (KIND == 3) {   // dependent LDS lookup chain (latency): idx -> lds -> idx
#pragma unroll
            for (int u = 0; u < 16; u++) a0 = lds[a0 & 2047];
        } else if (KIND == 4) {   // 8 independent LDS lookups per step (throughput)
#pragma unroll
            for (int u = 0; u < 2; u++) {
                a0 = lds[a0 & 2047]; a1 = lds[a1 & 2047]; a2 = lds[a2 & 2047]; a3 = lds[a3 & 2047];
                a4 = lds[a4 & 2047]; a5 = lds[a5 & 2047]; a6 = lds[a6 & 2047]; a7 = lds[a7 & 2047];
            }
        } else if (KIND == 5) {   // packed u16 ops
#pragma unroll
            for (int u = 0; u < 8; u++) {
                asm volatile("v_pk_sub_u16 %0, %0, %1 clamp" : "+v"(a0) : "v"(a1));
                asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(a2) : "v"(a3));
                asm volatile("v_pk_mad_u16 %0, %0, %1, %2" : "+v"(a4) : "v"(a5), "v"(a6));
                asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a7) : "v"(a1));
            }
        } else if (KIND == 6) {   // VALU + SALU interleaved (does scalar work issue beside vector work?)
            uint32_t s0 = __builtin_amdgcn_readfirstlane(a0);
#pragma unroll
            for (int u = 0; u < 8; u++) {
                a0 += a1; s0 = s0 * 3 + 1; a1 += a2; s0 ^= s0 >> 3; a2 += a3; s0 += 7; a3 += a4; s0 = s0 << 1 | 1;
            }
            a4 += s0;
        } else if (KIND == 7) {   // v_readlane + v_cmp -> sgpr mask + v_cndmask with sgpr mask (the select idiom)
#pragma unroll
            for (int u = 0; u < 8; u++) {
                a0 = (a1 > a2) ? a3 : a0; a1 = (a2 > a3) ? a4 : a1; a2 = (a3 > a4) ? a5 : a2; a3 = (a4 > a5) ? a6 : a3;
            }
        }
    }
    const long long t1 = clock64();
    out[blockIdx.x * 64 + lane] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
    if (lane == 0) ((long long *)(out + 1048576))[blockIdx.x] = t1 - t0;
}

template <int KIND>
int run(const char *name, int per_iter, uint32_t *d_out) {
    const int iters = 20000;
    for (int wpc : {4, 8, 16, 32}) {       // waves per CU (all four SIMDs): 1, 2, 4, 8 per SIMD
        const int grid = 256 * wpc;
        hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k_issue<KIND>, dim3(grid), dim3(64), 0, 0, d_out, 100, 1u);
        CHK(hipDeviceSynchronize());
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_issue<KIND>, dim3(grid), dim3(64), 0, 0, d_out, iters, 1u);
        CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<long long> cyc(grid);
        CHK(hipMemcpy(cyc.data(), d_out + 1048576, grid * sizeof(long long), hipMemcpyDeviceToHost));
        double avg = 0; for (long long c : cyc) avg += (double)c; avg /= grid;
        const double inst_per_wave = (double)iters * per_iter;
        // clock64 ticks at 100 MHz on gfx9 (s_memrealtime) or shader clock (s_memtime): report both views
        printf("%-28s waves/SIMD %d: %.3f ms; wall cycles@2.4GHz per wave-instr per SIMD %.2f; clock64 ticks per wave-instr (per wave) %.3f\n",
               name, wpc / 4, ms, ms * 1e-3 * 2.4e9 / (inst_per_wave * (wpc / 4.0)), avg / inst_per_wave);
    }
    return 0;
}

int main() {
    uint32_t *d_out; CHK(hipMalloc(&d_out, (1048576 + 2 * 8192 * 2) * 4));
    run<0>("v_add_u32 x8 indep", 64, d_out);
    run<1>("v_add_u32 dependent", 64, d_out);
    run<2>("int mix indep", 64 + 24, d_out);
    run<3>("lds lookup dependent", 32, d_out);
    run<4>("lds lookup x8 indep", 32, d_out);
    run<5>("v_pk_*_u16", 32, d_out);
    run<6>("valu+salu interleaved", 64, d_out);
    run<7>("cmp+cndmask", 64, d_out);
    return 0;
}
