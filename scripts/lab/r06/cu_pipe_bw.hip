// Lab probe (not part of the product): what one CU's vector-memory path sustains against HBM when the chip is NOT saturated
// (few workgroups) and when it is (one per CU), for the address shapes of the flash kernel's LDS-DMA requests and output stores.
//   hipcc --offload-arch=gfx950 -O3 cu_pipe_bw.hip -o bin/cu_pipe_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

// shape 0: 1 KiB contiguous per wave instruction
// shape 1: 4 rows x 256 B contiguous (row stride 2 KiB)
// shape 2: pair rows, one plane per instruction: 4 rows x 4 half lines of 64 B (128 B apart); the next instruction takes the
//          other halves (+64 B): what the kernel's k_hi / k_lo (q_hi / q_lo) requests and its hi / lo stores look like
// shape 3: 4 rows x 2 whole lines of 128 B (256 B apart): the same bytes as two instructions of shape 2, whole lines each
template <int SHAPE, bool STORE>
__global__ __launch_bounds__(256) void pipe_kernel(char* mem, size_t wg_stride, int n_instr, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char* base = mem + (size_t)blockIdx.x * wg_stride + (size_t)wave * (wg_stride / 4);
    const u32x4 val = {(unsigned)lane, 1u, 2u, 3u};
    // every wave walks its own quarter of the workgroup's region, 2 KiB rows
    for (int i = 0; i < n_instr; ++i) {
        size_t off;
        if (SHAPE == 0) off = (size_t)i * 1024 + lane * 16;
        else if (SHAPE == 1) off = (size_t)(i >> 3) * 8192 + (i & 7) * 256 + (size_t)(lane >> 4) * 2048 + (lane & 15) * 16;
        else if (SHAPE == 2) off = (size_t)(i >> 2) * 8192 + ((i >> 1) & 1) * 512 + (i & 1) * 64 + (size_t)(lane >> 4) * 2048 + ((lane & 15) >> 2) * 128 + (lane & 3) * 16;
        else off = (size_t)(i >> 2) * 8192 + (i & 3) * 512 + (size_t)(lane >> 4) * 2048 + ((lane & 15) >> 3) * 128 + (lane & 7) * 16;
        if (STORE) *reinterpret_cast<u32x4*>(base + off) = val;
        else __builtin_amdgcn_global_load_lds(base + off, (lds_ptr_t)(lds + wave * 16384 + (i & 15) * 1024), 16, 0, 0);
        if (!STORE && (i & 7) == 7) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");      // <= 32 requests of a wave in flight
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0) sink[blockIdx.x] = *reinterpret_cast<unsigned*>(lds + 64);
}

template <int SHAPE, bool STORE>
static void run(const char* name, char* mem, size_t wg_stride, int n_instr, unsigned* sink, int wgs) {
    auto k = pipe_kernel<SHAPE, STORE>;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(wgs), dim3(256), 65536, 0, mem, wg_stride, n_instr, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    const double bytes = (double)wgs * 4 * n_instr * 1024.0;
    printf("%-58s wgs %3d  %8.3f ms  %6.2f TB/s  %6.1f B/clk/CU @2.0GHz\n", name, wgs, ms, bytes / ms / 1e9, bytes / wgs / (ms * 1e-3 * 2.0e9));
    fflush(stdout);
}

int main() {
    const size_t wg_stride = (size_t)64 << 20;          // 64 MiB per workgroup: 16 MiB per wave
    const int wgs_max = 256;
    char* mem;
    if (hipMalloc(&mem, wg_stride * wgs_max) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(mem, 1, wg_stride * wgs_max);
    unsigned* sink;
    hipMalloc(&sink, 4096 * 4);
    const int n = 8192;                                 // 8 MiB per wave (every shape stays inside the wave's 16 MiB)
    for (int wgs : {8, 32, 256}) {
        run<0, false>("DMA   1 KiB contiguous", mem, wg_stride, n, sink, wgs);
        run<1, false>("DMA   4 rows x 256 B", mem, wg_stride, n, sink, wgs);
        run<2, false>("DMA   4 rows x 4 half lines (hi plane, then lo plane)", mem, wg_stride, n, sink, wgs);
        run<3, false>("DMA   4 rows x 2 whole lines", mem, wg_stride, n, sink, wgs);
        run<0, true>("store 1 KiB contiguous", mem, wg_stride, n, sink, wgs);
        run<1, true>("store 4 rows x 256 B", mem, wg_stride, n, sink, wgs);
        run<2, true>("store 4 rows x 4 half lines (hi, then lo)", mem, wg_stride, n, sink, wgs);
        run<3, true>("store 4 rows x 2 whole lines", mem, wg_stride, n, sink, wgs);
    }
    return 0;
}
