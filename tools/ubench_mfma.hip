// Micro-benchmarks that bound the sweep kernels' inner loop on MI355X (tuning aid, not part of the product).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma.hip -o build/ubench_mfma && build/ubench_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
// LDS-DMA through a buffer resource (MUBUF `buffer_load_dwordx4 ... lds`): per-lane 32-bit offset + scalar base instead of a
// 64-bit address per lane
__device__ __forceinline__ void blds16(__amdgpu_buffer_rsrc_t rsrc, int voff, void* l) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)l, 16, voff, 0, 0, 0);
}
__device__ const char* g_src;

template <int MODE>
__global__ __launch_bounds__(512, 2) void k(int iters, int* out, const char* src) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 49152 / 4; i += 512) ((int*)smem)[i] = i;
    __syncthreads();
    v16i acc0 = {0}, acc1 = {0};
    v4i a0 = {lane, 1, 2, 3}, a1 = {4, lane, 6, 7}, b0 = {1, 1, lane, 1}, b1 = {2, 2, 2, lane};
    v4i n_a0 = a0, n_a1 = a1, n_b0 = b0, n_b1 = b1, n_t0 = a0, n_t1 = a1;
    // swizzled like k_sweep4: row R = lane&31, logical chunk c = lane>>5 (+2 for the second K half) -> physical c ^ ((R>>2)&3)
    const int R = lane & 31, sw = (R >> 2) & 3, g = lane >> 5;
    const char* base = smem + R * 64 + (wid & 3) * 2048;
    const int o0 = (g ^ sw) << 4, o1 = ((2 + g) ^ sw) << 4;
    // like the sweep: 50 workgroups share one streaming tile and sit on the same XCD (block b -> XCD b%8)
    const int tlog = (blockIdx.x % 8) * 32 + blockIdx.x / 8;
    const char* cur = src + (size_t)((MODE & 64) ? blockIdx.x : tlog / 50) * 128 * 76800 + (size_t)(wid * 16 + (lane >> 2)) * 76800 + (lane & 3) * 16;
    if (MODE & 256) cur = src + (size_t)(tlog / 50) * 128 * 76800 + wid * 1024 + lane * 16;   // dense 8 KB tiles: 1 KB contiguous per instruction
    constexpr int CSTEP = (MODE & 256) ? 8192 : 64;
    int stg = 0;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, 0x7fffffff, 0x00020000);
    v4i r0 = {0}, r1 = {0}, r2 = {0};
    if (MODE & 128) { r0 = *(const v4i*)cur; cur += CSTEP; r1 = *(const v4i*)cur; cur += CSTEP; r2 = *(const v4i*)cur; cur += CSTEP; }
    // MODE & 512: streaming fragments straight from global memory into VGPRs (no LDS), 3 k-tiles in flight;
    // lane -> row (wid&3)*32 + (lane&31), 16-byte chunk lane>>5 (and +2), like the MFMA B-operand layout
    const char* curD = src + (size_t)(tlog / 50) * 128 * 76800 + (size_t)((wid & 3) * 32 + (lane & 31)) * 76800 + (lane >> 5) * 16;
    v4i d0a = {0}, d0b = {0}, d1a = {0}, d1b = {0}, d2a = {0}, d2b = {0};
    if (MODE & 512) {
        d0a = *(const v4i*)curD; d0b = *(const v4i*)(curD + 32); curD += 64;
        d1a = *(const v4i*)curD; d1b = *(const v4i*)(curD + 32); curD += 64;
        d2a = *(const v4i*)curD; d2b = *(const v4i*)(curD + 32); curD += 64;
    }
    if (MODE & 32) for (int i = 0; i < 5; ++i) { glds16(cur, smem + 16384 + 8192 + i * 8192 + wid * 1024); cur += CSTEP; }
    for (int it = 0; it < iters; ++it) {
        if (MODE & 32) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        if (MODE & 4) __builtin_amdgcn_s_barrier();
        if ((MODE & 32) && (MODE & 1024)) {   // only one wave per SIMD issues (two pieces each)
            if (wid < 4) { glds16(cur, smem + 16384 + 8192 + stg + wid * 2048); glds16(cur + 8 * 76800, smem + 16384 + 8192 + stg + wid * 2048 + 1024); }
            cur += CSTEP; stg = (stg + 8192 == 6 * 8192) ? 0 : stg + 8192; if ((it & 1023) == 1023) cur -= 1024 * CSTEP;
        } else if ((MODE & 32) && (MODE & 2048)) {   // four dword DMA instructions instead of one dwordx4
            for (int q = 0; q < 4; ++q)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(cur + q * 4),
                                                 (__attribute__((address_space(3))) void*)(smem + 16384 + 8192 + stg + wid * 1024 + q * 256), 4, 0, 0);
            cur += CSTEP; stg = (stg + 8192 == 6 * 8192) ? 0 : stg + 8192; if ((it & 1023) == 1023) cur -= 1024 * CSTEP;
        } else
        if ((MODE & 32) && (MODE & 32768)) { blds16(rsrc, (int)(cur - src), smem + 16384 + 8192 + stg + wid * 1024); cur += CSTEP; stg = (stg + 8192 == 6 * 8192) ? 0 : stg + 8192; if ((it & 1023) == 1023) cur -= 1024 * CSTEP; }
        else if (MODE & 32) { glds16(cur, smem + 16384 + 8192 + stg + wid * 1024); cur += CSTEP; stg = (stg + 8192 == 6 * 8192) ? 0 : stg + 8192; if ((it & 1023) == 1023) cur -= 1024 * CSTEP; }
        if (MODE & 128) {   // register staging: global_load_dwordx4 -> VGPR -> ds_write_b128, 3 loads in flight
            *reinterpret_cast<v4i*>(smem + 16384 + 8192 + stg + wid * 1024 + lane * 16) = r0;
            r0 = r1; r1 = r2; r2 = *(const v4i*)cur;
            cur += CSTEP; stg = (stg + 8192 == 6 * 8192) ? 0 : stg + 8192; if ((it & 1023) == 1023) cur -= 1024 * CSTEP;
        }
        if (MODE & 2) {   // 6 fragment reads per 4 MFMAs, like k_sweep4
            const char* st = base + ((it & 3) * 8192);
            if (MODE & 16) {   // software pipelined: fragments read now are used in the NEXT step
                n_a0 = *reinterpret_cast<const v4i*>(st + o0);
                if (!(MODE & 8192)) n_a1 = *reinterpret_cast<const v4i*>(st + 2048 + o0);
                if (!(MODE & 512)) n_b0 = *reinterpret_cast<const v4i*>(st + 16384 + o0);
                if (!(MODE & (4096 | 8192))) n_t0 = *reinterpret_cast<const v4i*>(st + o1);
                if (!(MODE & (4096 | 8192))) n_t1 = *reinterpret_cast<const v4i*>(st + 2048 + o1);
                if (!(MODE & (512 | 8192))) n_b1 = *reinterpret_cast<const v4i*>(st + 16384 + o1);
            } else {
                a0 = *reinterpret_cast<const v4i*>(st + o0);
                a1 = *reinterpret_cast<const v4i*>(st + 2048 + o0);
                b0 = *reinterpret_cast<const v4i*>(st + 16384 + o0);
                v4i t0 = *reinterpret_cast<const v4i*>(st + o1);
                v4i t1 = *reinterpret_cast<const v4i*>(st + 2048 + o1);
                b1 = *reinterpret_cast<const v4i*>(st + 16384 + o1);
                a0 += t0; a1 += t1;
            }
        }
        if (MODE & 16384) {   // no matrix work: one VALU op per fragment keeps the reads alive
            acc0[0] += a0[0] ^ b0[1]; acc1[1] += a1[2] ^ b1[3];
        } else {
        acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, b0, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b1, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, b1, acc1, 0, 0, 0);
        }
        if (MODE & 16) { a0 = n_a0 + n_t0; a1 = n_a1 + n_t1; b0 = n_b0; b1 = n_b1; }
        if (MODE & 512) {
            n_b0 = d0a; n_b1 = d0b; d0a = d1a; d0b = d1b; d1a = d2a; d1b = d2b;
            d2a = *(const v4i*)curD; d2b = *(const v4i*)(curD + 32);
            curD += 64; if ((it & 1023) == 1023) curD -= 65536;
        }
        if ((MODE & 8) && (it % 12) == 11) {   // an epilogue-like VALU burst every 12 steps
            float s = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) { float d = (float)acc0[r] * 0.5f - 1.0f; s += d * d; d = (float)acc1[r] * 0.25f - 2.0f; s += d * d; acc0[r] = 0; acc1[r] = 0; }
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
            if (lane == 0) ((float*)smem)[12000 + wid] += s;
        }
    }
    int r = 0;
    for (int i = 0; i < 16; ++i) r += acc0[i] + acc1[i];
    if (r == 0x7fffffff) out[threadIdx.x] = r;
}

template <int MODE> void run(const char* name, int wg_threads, size_t lds) {
    int* d; hipMalloc(&d, 4096);
    const int iters = 20000, grid = 256;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    static char* src = nullptr;
    if (!src) { hipMalloc(&src, (size_t)256 * 128 * 76800 + (1 << 20)); hipMemset(src, 1, (size_t)256 * 128 * 76800 + (1 << 20)); }
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(wg_threads), lds, 0, 100, d, src);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(wg_threads), lds, 0, iters, d, src);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double macs = (double)grid * (wg_threads / 64) * iters * 4.0 * 32768.0;
    printf("%-44s %2d waves/CU  %7.3f ms  %7.1f TOP/s  (%.1f cycles/step/SIMD @2.0GHz)\n", name, wg_threads / 64, ms, 2 * macs / ms / 1e9,
           ms * 1e-3 * 2.0e9 / iters);
    hipFree(d);
}

int main() {
    run<0>("mfma only", 512, 50000);
    run<0>("mfma only, 1 wave/SIMD", 256, 50000);
    run<2>("mfma + 6 ds_read_b128 per 4 mfma", 512, 50000);
    run<4>("mfma + barrier per step", 512, 50000);
    run<6>("mfma + reads + barrier", 512, 50000);
    run<8>("mfma + epilogue burst every 12 steps", 512, 50000);
    run<14>("mfma + reads + barrier + epilogue", 512, 50000);
    run<14>("same, 150 KB LDS", 512, 150000);
    run<18>("mfma + pipelined reads (no conflicts)", 512, 50000);
    run<22>("mfma + pipelined reads + barrier", 512, 50000);
    run<30>("mfma + pipelined reads + barrier + epilogue", 512, 50000);
    run<18>("mfma + pipelined reads, 1 wave/SIMD", 256, 50000);
    run<36>("mfma + barrier + LDS-DMA stream (8KB/step)", 512, 100000);
    run<54>("mfma + pipelined reads + barrier + LDS-DMA", 512, 100000);
    run<62>("mfma + pipelined reads + barrier + LDS-DMA + epi", 512, 100000);
    run<100>("mfma + barrier + LDS-DMA, unshared (HBM-bound)", 512, 100000);
    run<32>("mfma + LDS-DMA, no barrier", 512, 100000);
    run<36 + 256>("mfma + barrier + LDS-DMA dense source", 512, 100000);
    run<54 + 256>("mfma + pipelined reads + barrier + LDS-DMA dense", 512, 100000);
    run<22 + 512>("mfma + 4 pipelined reads + barrier + direct-to-VGPR stream", 512, 100000);
    run<30 + 512>("mfma + 4 pipelined reads + barrier + direct stream + epi", 512, 100000);
    run<18 + 512>("mfma + 4 pipelined reads + direct stream, no barrier", 512, 100000);
    run<50>("mfma + 6 pipelined reads + LDS-DMA, no barrier", 512, 100000);
    run<54 + 4096>("mfma + 4 pipelined reads + barrier + LDS-DMA", 512, 100000);
    run<54 + 8192>("mfma + 2 pipelined reads + barrier + LDS-DMA", 512, 100000);
    run<54 + 1024>("mfma + 6 reads + barrier + LDS-DMA by 4 waves", 512, 100000);
    run<54 + 2048>("mfma + 6 reads + barrier + LDS-DMA as 4 x b32", 512, 100000);
    run<22 + 4096>("mfma + 4 pipelined reads + barrier", 512, 100000);
    run<16384 + 18>("NO mfma: 6 pipelined reads only", 512, 100000);
    run<16384 + 18 + 4096>("NO mfma: 4 pipelined reads only", 512, 100000);
    run<16384 + 36>("NO mfma: barrier + LDS-DMA only", 512, 100000);
    run<16384 + 32>("NO mfma: LDS-DMA only, no barrier", 512, 100000);
    run<16384 + 54>("NO mfma: 6 reads + barrier + LDS-DMA", 512, 100000);
    run<16384 + 54 + 4096>("NO mfma: 4 reads + barrier + LDS-DMA", 512, 100000);
    run<16384 + 22>("NO mfma: 6 reads + barrier", 512, 100000);
    run<36 + 32768>("mfma + barrier + LDS-DMA via buffer_load lds", 512, 100000);
    run<54 + 32768>("mfma + pipelined reads + barrier + buffer_load lds", 512, 100000);
    run<16384 + 32 + 32768>("NO mfma: buffer_load lds only, no barrier", 512, 100000);
    run<4 + 128>("mfma + barrier + reg-staged stream", 512, 100000);
    run<22 + 128>("mfma + pipelined reads + barrier + reg-staged", 512, 100000);
    run<22 + 128 + 256>("mfma + pipelined reads + barrier + reg-staged dense", 512, 100000);
    run<30 + 128>("mfma + pipelined reads + barrier + reg-staged + epi", 512, 100000);
    return 0;
}
