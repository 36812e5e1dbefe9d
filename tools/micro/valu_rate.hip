// micro-benchmark: issue rate of scalar-FP32, packed-FP32 and v_readlane/v_cmp instructions on one MI355X
// (wave-instructions per SIMD per cycle).  Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define REP 256
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    f2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7};
    f2 A = {a, a}, B = {b, b};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < REP / 8; ++r) {
            if (MODE == 0) {        // 8 independent v_mul_f32 / v_add_f32 chains (no FMA contraction: separate ops)
                asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                             "v_add_f32 %4, %4, %9\n v_add_f32 %5, %5, %9\n v_add_f32 %6, %6, %9\n v_add_f32 %7, %7, %9\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
            } else if (MODE == 1) { // 8 packed ops on 4 register pairs
                asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %5\n v_pk_add_f32 %3, %3, %5\n"
                             "v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %5\n v_pk_add_f32 %3, %3, %5\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(A), "v"(B));
            } else if (MODE == 2) { // 8 v_fma_f32
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
            } else if (MODE == 3) { // 8 v_cmp + v_cndmask pairs -> 8 instructions
                asm volatile("v_cmp_lt_f32 vcc, %0, %4\n v_cndmask_b32 %0, %0, %5, vcc\n v_cmp_lt_f32 vcc, %1, %4\n v_cndmask_b32 %1, %1, %5, vcc\n"
                             "v_cmp_lt_f32 vcc, %2, %4\n v_cndmask_b32 %2, %2, %5, vcc\n v_cmp_lt_f32 vcc, %3, %4\n v_cndmask_b32 %3, %3, %5, vcc\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b) : "vcc");
            } else if (MODE == 4) { // 8 v_readlane_b32
                int s0, s1, s2, s3, s4, s5, s6, s7;
                asm volatile("v_readlane_b32 %0, %8, 3\n v_readlane_b32 %1, %9, 5\n v_readlane_b32 %2, %10, 7\n v_readlane_b32 %3, %11, 9\n"
                             "v_readlane_b32 %4, %8, 11\n v_readlane_b32 %5, %9, 13\n v_readlane_b32 %6, %10, 15\n v_readlane_b32 %7, %11, 17\n"
                             : "=s"(s0), "=s"(s1), "=s"(s2), "=s"(s3), "=s"(s4), "=s"(s5), "=s"(s6), "=s"(s7) : "v"(x0), "v"(x1), "v"(x2), "v"(x3));
                x4 += __int_as_float(s0 ^ s1 ^ s2 ^ s3 ^ s4 ^ s5 ^ s6 ^ s7) * 0.f;
            } else if (MODE == 5) { // 8 packed FMA
                asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                             "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(A), "v"(B));
            } else if (MODE == 6) { // 8 v_rcp_f32 (transcendental)
                asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
            } else if (MODE == 7) { // 8 v_div_scale/v_div_fmas/v_div_fixup mix (VOP3 with 3 sources)
                asm volatile("v_div_fixup_f32 %0, %0, %4, %5\n v_div_fixup_f32 %1, %1, %4, %5\n v_div_fixup_f32 %2, %2, %4, %5\n v_div_fixup_f32 %3, %3, %4, %5\n"
                             "v_div_fixup_f32 %0, %0, %4, %5\n v_div_fixup_f32 %1, %1, %4, %5\n v_div_fixup_f32 %2, %2, %4, %5\n v_div_fixup_f32 %3, %3, %4, %5\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b));
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}
template <int MODE>
void run(const char *name, float *d, int blocks) {
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(d, 10, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(d, iters, 1.0001f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double winstr = double(blocks) * 4 * double(iters) * REP;          // wave-instructions
    std::printf("%-28s blocks %5d  %8.3f ms  %8.2f G wave-instr/s  = %6.3f per SIMD per ns (1024 SIMDs)\n", name, blocks, ms, winstr / ms / 1e6, winstr / ms / 1e6 / 1024.0);
}
int main() {
    float *d; hipMalloc(&d, 256 * 8192 * sizeof(float));
    for (int blocks : {256 * 2, 256 * 8}) {   // 2 and 8 waves per SIMD
        run<0>("v_mul/v_add_f32", d, blocks);
        run<1>("v_pk_mul/v_pk_add_f32", d, blocks);
        run<2>("v_fma_f32", d, blocks);
        run<5>("v_pk_fma_f32", d, blocks);
        run<3>("v_cmp+v_cndmask", d, blocks);
        run<4>("v_readlane_b32", d, blocks);
        run<6>("v_rcp_f32", d, blocks);
        run<7>("v_div_fixup_f32", d, blocks);
    }
    return 0;
}
