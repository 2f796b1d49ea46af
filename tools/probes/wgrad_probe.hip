// Micro-probe for k_wgrad: one job shape at a time on synthetic stashes; prints time per (worker, item).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I../../include -I../../nvfi_amd/csrc -o wgrad_probe wgrad_probe.hip
#include "../../nvfi_amd/csrc/engine.hip"
#include <cstdarg>
#include <cstring>
#include <cstdio>
#include <cstdlib>
int nvfi_fail(int code, const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); return code; }
void prof_begin(int, hipStream_t) {}
void prof_end(int, hipStream_t) {}

static void run(const char* name, int a_regs, int b_regs, int bmode, int ntiles, int nslab, int njobs) {
    const size_t a_stride = (size_t)a_regs * REGF, b_stride = (size_t)b_regs * REGF;
    float *A, *B, *B2, *slabs; int* count;
    hipMalloc(&A, a_stride * ntiles * 4); hipMalloc(&B, b_stride * ntiles * 4); hipMalloc(&B2, b_stride * ntiles * 4);
    hipMemset(A, 0, a_stride * ntiles * 4); hipMemset(B, 0, b_stride * ntiles * 4); hipMemset(B2, 0, b_stride * ntiles * 4);
    const size_t slab = (size_t)(2 * a_regs) * (2 * b_regs) + 2 * a_regs;
    hipMalloc(&slabs, slab * nslab * njobs * 4);
    hipMalloc(&count, 4);
    int n = ntiles * 32; hipMemcpy(count, &n, 4, hipMemcpyHostToDevice);
    WgradJobs wj; wj.n = njobs;
    for (int j = 0; j < njobs; ++j) {
        WgradJob& J = wj.j[j]; std::memset((void*)&J, 0, sizeof(J));
        J.A = A; J.a_tile_stride = a_stride; J.a_regs = a_regs; J.B = B; J.B2 = B2; J.b_tile_stride = b_stride; J.b_regs = b_regs; J.bmode = bmode;
        J.count = count; J.cap_tiles = ntiles; J.nrep = 1; J.slabs = slabs + slab * nslab * j; J.nslab = nslab;
    }
    const bool tanm = bmode == BM_SILU_TAN || bmode == BM_RELU_TAN;
    const int mt = a_regs == 64 ? (tanm ? 2 : WGRAD_MT) : 1, ktw = b_regs == 16 ? 1 : 2;
    const int wps = ((a_regs >> 4) / mt) * ((b_regs >> 4) / ktw);
    const int nworkers = nslab * wps;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_wgrad, dim3((nworkers + 3) / 4, njobs), dim3(WG_THREADS), 0, 0, wj);
        hipEventRecord(e1);
        hipDeviceSynchronize();
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double items = (double)ntiles / nslab;       // per worker
    const int mfma_per_item = mt * ktw * 16;
    printf("%-26s workers=%5d (x%d jobs) items/worker=%6.1f  %.3f ms  %.2f us/item  (MFMA floor %.2f us/item)\n", name, nworkers, njobs, items, ms,
           ms * 1e3 / items, mfma_per_item * 64 / 2400.0);
    hipFree(A); hipFree(B); hipFree(B2); hipFree(slabs); hipFree(count);
}

int main(int argc, char** argv) {
    const int only = argc > 1 ? atoi(argv[1]) : -1;   // run one configuration (for counter collection)
    const int mul = argc > 2 ? atoi(argv[2]) : 1;     // multiply the items per worker
    int k = 0;
#define RUN(...) do { if (only < 0 || only == k) run(__VA_ARGS__); ++k; } while (0)
    // one resident round: 1024 waves
    RUN("64x64 SILU, 1 job", 64, 64, BM_SILU, 512 * 64 * mul, 512, 1);
    RUN("64x64 RAW,  1 job", 64, 64, BM_RAW, 512 * 64 * mul, 512, 1);
    RUN("64x64 SILU, 4 jobs x128", 64, 64, BM_SILU, 128 * 64 * mul, 128, 4);
    RUN("64x64 SILU_TAN, 1 job", 64, 64, BM_SILU_TAN, 256 * 64 * mul, 256, 1);
    RUN("16x64 SILU, 1 job", 16, 64, BM_SILU, 512 * 64 * mul, 512, 1);
    RUN("64x16 RAW, 1 job", 64, 16, BM_RAW, 1024 * 64 * mul, 1024, 1);
    RUN("64x64 SILU, 2 rounds", 64, 64, BM_SILU, 1024 * 32 * mul, 1024, 1);
    RUN("64x64 SILU, half round", 64, 64, BM_SILU, 256 * 64 * mul, 256, 1);
    return 0;
}
