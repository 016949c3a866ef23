// Developer micro-benchmark (round 4): the sliding dot product of phase 3b in packed fp32 (v_pk_fma_f32, sample pairs read
// with ds_read_b64 at 4-byte-aligned addresses, wave-uniform taps in SGPR pairs) against the shipped fp64 form, on random
// live units at the occupancy of config 2 (two 512-thread workgroups per CU, 80 KB of LDS each).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/pk_dot tools/micro/pk_dot.hip && /tmp/pk_dot
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#include <cstdlib>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(4))) float* const_f32_ptr;
typedef const __attribute__((address_space(4))) double* const_f64_ptr;
constexpr int kR = 5, kU = 8, kM = 4864;
__device__ __forceinline__ void load_pairs(unsigned addr, f32x2 (&x)[4]) {
    asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:8\n\tds_read_b64 %2, %4 offset:16\n\tds_read_b64 %3, %4 offset:24\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3]) : "v"(addr) : "memory");
}
__device__ __forceinline__ void load8(unsigned addr, double (&x)[8]) {
    asm volatile("ds_read_b64 %0, %8\n\tds_read_b64 %1, %8 offset:8\n\tds_read_b64 %2, %8 offset:16\n\tds_read_b64 %3, %8 offset:24\n\t"
                 "ds_read_b64 %4, %8 offset:32\n\tds_read_b64 %5, %8 offset:40\n\tds_read_b64 %6, %8 offset:48\n\tds_read_b64 %7, %8 offset:56\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3]), "=&v"(x[4]), "=&v"(x[5]), "=&v"(x[6]), "=&v"(x[7]) : "v"(addr) : "memory");
}
__device__ __forceinline__ void dot32(unsigned eh_addr, const_f32_ptr q, int L, float (&B)[kR]) {
    constexpr int S = kR - 1;
    f32x2 acc[kR];
#pragma unroll
    for (int r = 0; r < kR; ++r) acc[r] = f32x2{0.f, 0.f};
    for (int t0 = 0; t0 < L + S; t0 += kU) {
        const const_f32_ptr qa = q + (t0 - S);
        const const_f32_ptr qb = q + (t0 - S - 1);
        float ta[kU + S], tb[kU + S + 2];
#pragma unroll
        for (int m = 0; m < kU + S; ++m) ta[m] = qa[m];
#pragma unroll
        for (int m = 0; m < kU + S + 2; ++m) tb[m] = qb[m];
        f32x2 x[4];
        load_pairs(eh_addr + 4u * (unsigned)t0, x);
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int r = 0; r < kR; ++r) {
                f32x2 tap;
                if ((r & 1) == 0) { const int m = 2 * k - r + S; tap = f32x2{ta[m], ta[m + 1]}; }
                else { const int m = 2 * k - r + S + 1; tap = f32x2{tb[m], tb[m + 1]}; }
                acc[r] = __builtin_elementwise_fma(tap, x[k], acc[r]);
            }
    }
#pragma unroll
    for (int r = 0; r < kR; ++r) B[r] = acc[r].x + acc[r].y;
}
__device__ __forceinline__ void dot64(unsigned e_addr, const_f64_ptr q, int L, double (&B)[kR]) {
    constexpr int S = kR - 1;
    for (int t0 = 0; t0 < L + S; t0 += kU) {
        const const_f64_ptr qs = q + (t0 - S);
        double taps[kU + S];
#pragma unroll
        for (int m = 0; m < kU + S; ++m) taps[m] = qs[m];
        double x[kU];
        load8(e_addr + 8u * (unsigned)t0, x);
#pragma unroll
        for (int u = 0; u < kU; ++u)
#pragma unroll
            for (int r = 0; r < kR; ++r) B[r] = fma(taps[u + S - r], x[u], B[r]);
    }
}
__device__ __forceinline__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
template <bool F32>
__global__ __launch_bounds__(512) void bench(const float* q32, const double* q64, const double* e, double* out, int L, int iters, int n_units, int align) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    double* ed = reinterpret_cast<double*>(sm);
    float* ef = reinterpret_cast<float*>(sm);
    if (F32) { for (int i = threadIdx.x; i < kM; i += blockDim.x) { const double v = e[i]; const float h = (float)v; ef[i] = h; ef[kM + i] = (float)(v - (double)h); } }
    else { for (int i = threadIdx.x; i < kM; i += blockDim.x) ed[i] = e[i]; }
    __syncthreads();
    const unsigned base = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) unsigned char*)sm;
    double total[kR] = {0, 0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        // a batch: 64 live units of one row in ascending order with random gaps (~14 % live)
        const unsigned wave = threadIdx.x / 64, lane = threadIdx.x & 63;
        unsigned u = (hash(it * 977u + wave * 131u + blockIdx.x) % 200u) + lane * 7u + (hash(it * 64u + lane + wave * 7919u) % 7u);
        u %= (unsigned)n_units;
        const int b = ((int)u * kR) & ~align;
        if (F32) {
            float B[kR];
            dot32(base + 4u * (unsigned)b, (const_f32_ptr)q32 + 64, L, B);
#pragma unroll
            for (int r = 0; r < kR; ++r) total[r] += (double)B[r];
        } else {
            double B[kR] = {0, 0, 0, 0, 0};
            dot64(base + 8u * (unsigned)b, (const_f64_ptr)q64 + 64, L, B);
#pragma unroll
            for (int r = 0; r < kR; ++r) total[r] += B[r];
        }
    }
    double s = 0; for (int r = 0; r < kR; ++r) s += total[r] * (r + 1);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main(int argc, char** argv) {
    const int align = argc > 1 ? atoi(argv[1]) : 0;
    const int L = 48, iters = 2000, blocks = 512, n_units = (kM - 128) / kR;
    std::vector<float> q32(256, 0.f); std::vector<double> q64(256, 0.0), e(kM);
    for (int j = 0; j < L; ++j) { q64[64 + j] = 0.5 + 0.5 * std::sin(3.14159 * (j + 0.5) / L); q32[64 + j] = (float)q64[64 + j]; }
    srand(1); for (auto& v : e) v = 1e-4 * ((rand() % 2001) - 1000) / 1000.0;
    float* dq32; double *dq64, *de, *dout;
    hipMalloc(&dq32, 1024); hipMalloc(&dq64, 2048); hipMalloc(&de, kM * 8); hipMalloc(&dout, blocks * 512 * 8);
    hipMemcpy(dq32, q32.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(dq64, q64.data(), 2048, hipMemcpyHostToDevice);
    hipMemcpy(de, e.data(), kM * 8, hipMemcpyHostToDevice);
    const size_t lds = 80 * 1024;
    hipFuncSetAttribute((const void*)bench<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void*)bench<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    std::vector<double> o32(blocks * 512), o64(blocks * 512);
    for (int rep = 0; rep < 3; ++rep) {
        float ms32, ms64;
        hipEventRecord(a); bench<false><<<blocks, 512, lds>>>(dq32, dq64, de, dout, L, iters, n_units, align); hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms64, a, b);
        hipMemcpy(o64.data(), dout, o64.size() * 8, hipMemcpyDeviceToHost);
        hipEventRecord(a); bench<true><<<blocks, 512, lds>>>(dq32, dq64, de, dout, L, iters, n_units, align); hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms32, a, b);
        hipMemcpy(o32.data(), dout, o32.size() * 8, hipMemcpyDeviceToHost);
        double worst = 0, scale = 0; for (size_t i = 0; i < o32.size(); ++i) { worst = std::fmax(worst, std::fabs(o32[i] - o64[i])); scale = std::fmax(scale, std::fabs(o64[i])); }
        printf("align %d: fp64 %.3f ms, packed fp32 %.3f ms (x%.2f); max |diff| %.3e of %.3e; err %s\n", align, ms64, ms32, ms64 / ms32, worst, scale, hipGetErrorString(hipGetLastError()));
    }
    return 0;
}
