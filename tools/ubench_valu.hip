// Micro-benchmark: issue cost of the VALU instructions the path tracer leans on (gfx950).  Each kernel runs a long chain of one
// instruction kind in every lane of enough waves to fill the chip; cost = time * SIMDs * clock / wave-instructions, in cycles
// per wave64 instruction per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o gpurun_out/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr int ITER = 4096;
#define KERNEL(name, T, init, body)                                         \
    __global__ void name(T *out) {                                          \
        T a = (T)(threadIdx.x + 1), b = init, c = (T)3, d = (T)5;            \
        T x0 = a, x1 = a + (T)1, x2 = a + (T)2, x3 = a + (T)3;               \
        for (int i = 0; i < ITER; ++i) {                                    \
            body(x0) body(x1) body(x2) body(x3)                             \
        }                                                                   \
        out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + c + d; \
    }
#define B_FMA32(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define B_MULHI(x) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x) : "v"(b));
#define B_MULLO(x) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(b));
#define B_ADD32(x) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(b));
#define B_RCP(x) asm volatile("v_rcp_f32 %0, %0" : "+v"(x));
#define B_SQRT(x) asm volatile("v_sqrt_f32 %0, %0" : "+v"(x));
#define B_FMA64(x) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define B_ADD64(x) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x) : "v"(b));
#define B_MUL64(x) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(b));
#define B_RCP64(x) asm volatile("v_rcp_f64 %0, %0" : "+v"(x));
#define B_CVT(x) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(x));
#define B_MIN3(x) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define B_XOR(x) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x) : "v"(b));
#define B_PKFMA(x) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
KERNEL(k_fma32, float, 1.0001f, B_FMA32)
KERNEL(k_add32, float, 1.0001f, B_ADD32)
KERNEL(k_mulhi, uint32_t, 0xD2511F53u, B_MULHI)
KERNEL(k_mullo, uint32_t, 0xD2511F53u, B_MULLO)
KERNEL(k_xor, uint32_t, 0xD2511F53u, B_XOR)
KERNEL(k_rcp, float, 1.0001f, B_RCP)
KERNEL(k_sqrt, float, 1.0001f, B_SQRT)
KERNEL(k_cvt, float, 1.0001f, B_CVT)
KERNEL(k_min3, float, 1.0001f, B_MIN3)
KERNEL(k_fma64, double, 1.0001, B_FMA64)
KERNEL(k_add64, double, 1.0001, B_ADD64)
KERNEL(k_mul64, double, 1.0001, B_MUL64)
KERNEL(k_rcp64, double, 1.0001, B_RCP64)
KERNEL(k_pkfma, double, 1.0001, B_PKFMA)

__global__ void k_mad64(unsigned long long *out) {
    unsigned long long x0 = threadIdx.x + 1, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    uint32_t b = 0xD2511F53u;
    for (int i = 0; i < ITER; ++i) {
        asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(x0) : "v"((uint32_t)x0), "v"(b) : "vcc");
        asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(x1) : "v"((uint32_t)x1), "v"(b) : "vcc");
        asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(x2) : "v"((uint32_t)x2), "v"(b) : "vcc");
        asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(x3) : "v"((uint32_t)x3), "v"(b) : "vcc");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3;
}

template <typename T, typename K>
void run(const char *name, K kernel, int clock_mhz, int cus) {
    const int blocks = cus * 16, threads = 256;  // 4 waves per SIMD
    T *out;
    hipMalloc(&out, sizeof(T) * blocks * threads);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), 0, 0, out);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), 0, 0, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double wave_instr = (double)blocks * (threads / 64) * ITER * 4;
    const double cycles = ms * 1e-3 * clock_mhz * 1e6 * (cus * 4);
    printf("%-10s %8.3f ms  %6.2f cycles per wave64 instruction per SIMD\n", name, ms, cycles / wave_instr);
    hipFree(out);
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int mhz = p.clockRate / 1000, cus = p.multiProcessorCount;
    printf("%s: %d CUs, %d MHz\n", p.name, cus, mhz);
    run<float>("fma_f32", k_fma32, mhz, cus);
    run<float>("add_f32", k_add32, mhz, cus);
    run<float>("min3_f32", k_min3, mhz, cus);
    run<float>("cvt_f32_u32", k_cvt, mhz, cus);
    run<uint32_t>("xor_b32", k_xor, mhz, cus);
    run<uint32_t>("mul_hi_u32", k_mulhi, mhz, cus);
    run<uint32_t>("mul_lo_u32", k_mullo, mhz, cus);
    run<unsigned long long>("mad_u64_u32", k_mad64, mhz, cus);
    run<float>("rcp_f32", k_rcp, mhz, cus);
    run<float>("sqrt_f32", k_sqrt, mhz, cus);
    run<double>("fma_f64", k_fma64, mhz, cus);
    run<double>("add_f64", k_add64, mhz, cus);
    run<double>("mul_f64", k_mul64, mhz, cus);
    run<double>("rcp_f64", k_rcp64, mhz, cus);
    run<double>("pk_fma_f32", k_pkfma, mhz, cus);
    return 0;
}
