// Micro-benchmark: what does ONE workgroup of 1024 threads move per cycle between its private HBM region and LDS,
// alone on the chip and with every other CU doing the same?  (developer tool; hipcc --offload-arch=gfx950 -O3)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double f64x2 __attribute__((ext_vector_type(2)));
constexpr int kChunk = 16384;   // doubles per round (128 KiB of LDS)

template <int MODE, bool NT>
__global__ __launch_bounds__(1024) void k(double* base, long long region, int rounds_per_pass, int passes, unsigned long long* cycles, double* sink) {
    extern __shared__ double lds[];
    double* reg = base + (long long)blockIdx.x * region;
    const int tid = threadIdx.x, nt = blockDim.x;
    double acc = 0;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int p = 0; p < passes; ++p)
        for (int r = 0; r < rounds_per_pass; ++r) {
            double* g = reg + (long long)r * kChunk;
            if (MODE == 0) {   // read, 8-byte loads, 8 in flight
#pragma unroll
                for (int e = 0; e < 16; e += 8) {
                    double v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = NT ? __builtin_nontemporal_load(g + tid + (e + j) * nt) : g[tid + (e + j) * nt];
#pragma unroll
                    for (int j = 0; j < 8; ++j) lds[tid + (e + j) * nt] = v[j];
                }
            } else if (MODE == 1) {   // read, 16-byte loads, 8 in flight (all of the chunk)
                f64x2 v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = NT ? __builtin_nontemporal_load(reinterpret_cast<f64x2*>(g) + tid + j * nt) : reinterpret_cast<f64x2*>(g)[tid + j * nt];
#pragma unroll
                for (int j = 0; j < 8; ++j) reinterpret_cast<f64x2*>(lds)[tid + j * nt] = v[j];
            } else if (MODE == 2) {   // read by LDS-direct loads
                const unsigned int dst0 = (unsigned int)(uintptr_t)(__attribute__((address_space(3))) double*)lds;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const unsigned int dst = (unsigned int)__builtin_amdgcn_readfirstlane((int)(dst0 + 16u * (unsigned int)(j * nt + (tid / 64) * 64)));
                    const f64x2* src = reinterpret_cast<f64x2*>(g) + tid + j * nt;
                    unsigned int m0s;
                    if (NT) asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0" : "=&s"(m0s) : "v"(src), "s"(dst) : "memory");
                    else asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(m0s) : "v"(src), "s"(dst) : "memory");
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else if (MODE == 3) {   // write, 16-byte stores
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    f64x2 v = reinterpret_cast<f64x2*>(lds)[tid + j * nt];
                    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<f64x2*>(g) + tid + j * nt); else reinterpret_cast<f64x2*>(g)[tid + j * nt] = v;
                }
            } else if (MODE == 4) {   // write, 8-byte stores
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    double v = lds[tid + j * nt];
                    if (NT) __builtin_nontemporal_store(v, g + tid + j * nt); else g[tid + j * nt] = v;
                }
            }
            __syncthreads();
            acc += lds[(tid * 7 + r) & (kChunk - 1)];
        }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (tid == 0) cycles[blockIdx.x] = t1 - t0;
    if (acc == 1.2345e-300) sink[0] = acc;
}

template <int MODE, bool NT>
void run(const char* name, int blocks, double* d, long long region, unsigned long long* d_cyc, double* d_sink) {
    const int rounds = (int)(region / kChunk), passes = 4;
    hipFuncSetAttribute((const void*)k<MODE, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, kChunk * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE, NT><<<blocks, 1024, kChunk * 8>>>(d, region, rounds, 1, d_cyc, d_sink);
    hipEventRecord(e0);
    k<MODE, NT><<<blocks, 1024, kChunk * 8>>>(d, region, rounds, passes, d_cyc, d_sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> c(blocks);
    hipMemcpy(c.data(), d_cyc, blocks * 8, hipMemcpyDeviceToHost);
    double mean = 0; for (auto v : c) mean += (double)v; mean /= blocks;
    const double bytes = (double)region * 8 * passes;
    printf("%-28s blocks %3d  %7.2f B/cycle/WG  (readcyclecounter)  %8.1f GB/s aggregate  %.3f ms\n", name, blocks, bytes / mean, bytes * blocks / ms * 1e-6, ms);
}

int main() {
    const long long region = 8 * kChunk;   // 1 MiB per workgroup
    double* d; hipMalloc(&d, 256 * region * 8); hipMemset(d, 0, 256 * region * 8);
    unsigned long long* d_cyc; hipMalloc(&d_cyc, 256 * 8);
    double* d_sink; hipMalloc(&d_sink, 8);
    for (int blocks : {256, 64, 8}) {
        run<0, false>("read 8B x8", blocks, d, region, d_cyc, d_sink);
        run<0, true>("read 8B x8 nt", blocks, d, region, d_cyc, d_sink);
        run<1, false>("read 16B x8", blocks, d, region, d_cyc, d_sink);
        run<1, true>("read 16B x8 nt", blocks, d, region, d_cyc, d_sink);
        run<2, false>("read lds-direct 16B x8", blocks, d, region, d_cyc, d_sink);
        run<2, true>("read lds-direct 16B x8 nt", blocks, d, region, d_cyc, d_sink);
        run<3, false>("write 16B", blocks, d, region, d_cyc, d_sink);
        run<3, true>("write 16B nt", blocks, d, region, d_cyc, d_sink);
        run<4, false>("write 8B", blocks, d, region, d_cyc, d_sink);
        run<4, true>("write 8B nt", blocks, d, region, d_cyc, d_sink);
    }
    return 0;
}
