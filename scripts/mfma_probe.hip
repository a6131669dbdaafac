// mfma_probe.hip — component ladder of the fp32 implicit-GEMM K loop on gfx950 (tuning tool, not part of libpfk).
//
//   /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/mfma_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
//
// One 256-thread block = four waves, each running the K-step of conv_gemm_v3_kernel's 64x64 tile: 16 dependent
// v_mfma_f32_32x32x2_f32 on one accumulator.  Around them the step's other work is switched on piece by piece:
//   BAR  one workgroup barrier per step
//   FR   the 8 fragment ds_read_b128 (swizzled 48 KB layout) that feed the MFMAs
//   ST   the 4 ds_write_b128 of the staged operands
//   LD   the 4 buffer_load_dwordx4 into registers (register staging: LD + ST = what the product kernel does)
//   DMA  the same 4 loads as buffer_load_dwordx4 ... lds (no staging registers, no ds_write), counted vmcnt + raw s_barrier
// with 1..3 blocks per CU, operands from an L2-resident or an HBM-sized array.  Prints MFMA-pipe utilisation per variant:
// the number the product kernel's 0.80 has to be read against.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BAR = 1, FR = 2, ST = 4, LD = 8, DMA = 16, FR4 = 32, AG = 64;   // AG: accumulator in AccVGPRs (inline asm)
//   // FR4: four fragment sets, each re-read for the NEXT step right after its last use
constexpr int STAGE = 128 * 32;   // floats: 64 A rows + 64 B rows of 32

// shader-clock cycles (s_memtime) and 100 MHz wall ticks (s_memrealtime) summed over the blocks of a launch: their ratio is the
// shader clock the chip actually sustained under this loop body
__device__ unsigned long long g_clk[2];

template <int MODE, int BPC>
__global__ __launch_bounds__(256, BPC) void probe(const float* __restrict__ src, unsigned src_bytes_mask, float* out, int steps,
                                                  unsigned row_stride /* bytes between the 32 rows a piece covers; 128 = one contiguous 4 KB piece */) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r0 = tid >> 3;
  const int key = (r0 >> 1) & 7;
  const int scol = (((tid & 7) ^ key)) << 2;                 // register staging: swizzled LDS column, linear global column
  const int gcol = scol;                                      // DMA: linear LDS position, swizzled global column (same involution)
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, 0x7fffffff, 0x00020000);
  // fragment offsets as in pfk_gemm.hip (frag_offsets<32>)
  const int wm0 = (wid >> 1) * 32, wn0 = (wid & 1) * 32;
  const int foff_a = (wm0 + (lane & 31)) * 32, foff_b = 64 * 32 + (wn0 + (lane & 31)) * 32;
  int ko[4];
  {
    const int hl = lane >> 5, k2 = ((lane & 31) >> 1) & 7;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) ko[kk] = ((kk * 2 + hl) ^ k2) << 2;
  }
  for (int i = tid; i < 3 * STAGE; i += 256) smem[i] = 0.001f * (float)(i & 63);
  __syncthreads();
  const unsigned long long c0 = clock64(), w0 = wall_clock64();
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  f32x4 fa0 = {1.f, 2.f, 3.f, 4.f}, fb0 = {0.5f, 0.25f, 0.125f, 1.f}, fa1 = fa0, fb1 = fb0;
  f32x4 ra[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) ra[q] = f32x4{(float)tid, 1.f, 2.f, 3.f};
  float* s_cur = smem;
  float* s_nxt = smem + STAGE;
  float* s_fill = smem + 2 * STAGE;
  // per-thread global byte offset of its 16 bytes in a 16 KB K-step slab; slabs walk through the source array
  const unsigned goff_reg = (unsigned)r0 * row_stride + (unsigned)(tid & 7) * 16u;
  const unsigned goff_dma = (unsigned)r0 * row_stride + (unsigned)gcol * 4u;
  unsigned slab = (unsigned)(blockIdx.x * 7919u) * 16384u;
  if constexpr ((MODE & FR) != 0) {
    fa0 = *reinterpret_cast<const f32x4*>(s_cur + foff_a + ko[0]);
    fb0 = *reinterpret_cast<const f32x4*>(s_cur + foff_b + ko[0]);
  }
  f32x4 ga[4], gb[4];
  if constexpr ((MODE & FR4) != 0) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      ga[kk] = *reinterpret_cast<const f32x4*>(s_cur + foff_a + ko[kk]);
      gb[kk] = *reinterpret_cast<const f32x4*>(s_cur + foff_b + ko[kk]);
    }
  }
  for (int j = 0; j < steps; ++j) {
    const unsigned sbase = slab & src_bytes_mask;
    if constexpr ((MODE & FR4) != 0) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[kk][s], gb[kk][s], acc, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          if (kk == 1) {
            const int q = s;
            if constexpr ((MODE & ST) != 0) *reinterpret_cast<f32x4*>(s_fill + (r0 + 32 * q) * 32 + scol) = ra[q];
            if constexpr ((MODE & LD) != 0)
              ra[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, goff_reg + (unsigned)q * 32u * row_stride, sbase, 0));
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        // set kk is free now: fetch sub-step kk of the NEXT step (its stage was completed a barrier ago) — 12 MFMAs ahead of its use
        ga[kk] = *reinterpret_cast<const f32x4*>(s_nxt + foff_a + ko[kk]);
        gb[kk] = *reinterpret_cast<const f32x4*>(s_nxt + foff_b + ko[kk]);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        if constexpr ((MODE & AG) != 0) {
          if ((kk & 1) == 0) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(fa0[s]), "v"(fb0[s]));
          else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(fa1[s]), "v"(fb1[s]));
        } else {
          if ((kk & 1) == 0) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[s], fb0[s], acc, 0, 0, 0);
          else acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[s], fb1[s], acc, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (s == 0 && (MODE & FR)) {     // next sub-step's fragments right after the first MFMA of this one
          const float* base = kk < 3 ? s_cur : s_nxt;
          const int k2 = kk < 3 ? kk + 1 : 0;
          if ((kk & 1) == 0) {
            fa1 = *reinterpret_cast<const f32x4*>(base + foff_a + ko[k2]);
            fb1 = *reinterpret_cast<const f32x4*>(base + foff_b + ko[k2]);
          } else {
            fa0 = *reinterpret_cast<const f32x4*>(base + foff_a + ko[k2]);
            fb0 = *reinterpret_cast<const f32x4*>(base + foff_b + ko[k2]);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        if (kk == 1) {                    // fillers: piece q = s in the second sub-step's four gaps
          const int q = s;
          if constexpr ((MODE & DMA) != 0) {
            // wave-uniform LDS base of this wave's 8 rows of piece q; each lane lands at base + lane * 16
            float* dst = s_fill + (q * 32 + wid * 8) * 32;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)dst, 16,
                                                     goff_dma + (unsigned)q * 32u * row_stride, sbase, 0, 0);
          } else {
            if constexpr ((MODE & ST) != 0) *reinterpret_cast<f32x4*>(s_fill + (r0 + 32 * q) * 32 + scol) = ra[q];
            if constexpr ((MODE & LD) != 0)
              ra[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, goff_reg + (unsigned)q * 32u * row_stride, sbase, 0));
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    }
    slab += 16384u;
    float* t = s_cur; s_cur = s_nxt; s_nxt = s_fill; s_fill = t;
    if constexpr ((MODE & DMA) != 0) {
      // the loads issued in THIS step may stay in flight; those of the previous step fill the stage that is read next
      asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
      if constexpr ((MODE & BAR) != 0) __builtin_amdgcn_s_barrier();
    } else if constexpr ((MODE & BAR) != 0) {
      __syncthreads();
    }
  }
  // keep everything alive
  if constexpr ((MODE & AG) != 0) asm volatile("s_nop 15\n s_nop 15" ::: "memory");   // MFMA -> accvgpr read hazard is ours with inline asm
  float v = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) v += acc[r];
#pragma unroll
  for (int q = 0; q < 4; ++q) v += ra[q][0];
  if (tid == 0) {
    atomicAdd(&g_clk[0], clock64() - c0);
    atomicAdd(&g_clk[1], wall_clock64() - w0);
  }
  if (v == 123.456f) out[blockIdx.x * 256 + tid] = v + s_cur[tid];
}

// bf16 matrix pipe alone: four independent accumulators per wave, v_mfma_f32_32x32x16_bf16 back to back (no LDS, no memory) — the
// sustained rate and shader clock the split-bf16 kernels (K8) have to be read against
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int BPC>
__global__ __launch_bounds__(256, BPC) void probe_bf16(float* out, int steps) {
  const unsigned long long c0 = clock64(), w0 = wall_clock64();
  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 a, b;
#pragma unroll
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * (float)(threadIdx.x + e)); b[e] = (__bf16)(0.002f * (float)(e + 1)); }
  for (int j = 0; j < steps; ++j) {
#pragma unroll
    for (int k = 0; k < 12; ++k) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
  }
  if (threadIdx.x == 0) {
    atomicAdd(&g_clk[0], clock64() - c0);
    atomicAdd(&g_clk[1], wall_clock64() - w0);
  }
  float v = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) v += acc[i][r];
  if (v == 123.456f) out[blockIdx.x * 256 + threadIdx.x] = v;
}

template <int BPC>
void run_bf16(float* out, double sustain_s) {
  const int grid = 256 * BPC, steps = 400;
  auto kern = probe_bf16<BPC>;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const auto t0 = std::chrono::steady_clock::now();
  do {
    for (int i = 0; i < 8; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, steps);
    hipDeviceSynchronize();
  } while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < sustain_s);
  unsigned long long zero[2] = {0, 0};
  hipMemcpyToSymbol(HIP_SYMBOL(g_clk), zero, sizeof zero);
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, steps);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  unsigned long long clk[2];
  hipMemcpyFromSymbol(clk, HIP_SYMBOL(g_clk), sizeof clk);
  const double mfmas = (double)grid * 4 * steps * 48;
  printf("  bf16 MFMA chain x%d blocks/CU, %s: %7.1f TFLOP/s (nominal dense peak 2516)  %7.1f MHz\n", BPC, sustain_s > 0 ? "sustained" : "cold     ",
         mfmas * 32768.0 / (best * 1e-3) / 1e12, clk[1] ? 100.0 * (double)clk[0] / (double)clk[1] : 0.0);
}

unsigned g_row_stride = 128;
double g_sustain_s = 0.0;        // > 0: repeat the launch for this long before the timed ones (lets power management settle)
double g_last_mhz = 0.0;
template <int MODE, int BPC>
double run(const float* src, unsigned mask, float* out, int steps) {
  auto kern = probe<MODE, BPC>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * STAGE * 4);
  const int grid = 256 * BPC;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 3 * STAGE * 4, 0, src, mask, out, steps / 4, g_row_stride);
  hipDeviceSynchronize();
  if (g_sustain_s > 0.0) {
    const auto t0 = std::chrono::steady_clock::now();
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < g_sustain_s) {
      for (int i = 0; i < 8; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 3 * STAGE * 4, 0, src, mask, out, steps, g_row_stride);
      hipDeviceSynchronize();
    }
  }
  unsigned long long zero[2] = {0, 0};
  hipMemcpyToSymbol(HIP_SYMBOL(g_clk), zero, sizeof zero);
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 3 * STAGE * 4, 0, src, mask, out, steps, g_row_stride);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  unsigned long long clk[2];
  hipMemcpyFromSymbol(clk, HIP_SYMBOL(g_clk), sizeof clk);
  g_last_mhz = clk[1] ? 100.0 * (double)clk[0] / (double)clk[1] : 0.0;
  const double mfmas = (double)grid * 4 * steps * 16;
  const double tf = mfmas * 4096.0 / (best * 1e-3) / 1e12;
  return tf;
}

template <int MODE>
void row(const char* name, const float* l2, unsigned l2mask, const float* hbm, unsigned hbmmask, float* out, int steps) {
  printf("%-34s", name);
  printf(" | L2-resident: x1 %6.1f  x2 %6.1f  x3 %6.1f", run<MODE, 1>(l2, l2mask, out, steps), run<MODE, 2>(l2, l2mask, out, steps),
         run<MODE, 3>(l2, l2mask, out, steps));
  if (MODE & (LD | DMA))
    printf(" | HBM-sized: x1 %6.1f  x2 %6.1f  x3 %6.1f", run<MODE, 1>(hbm, hbmmask, out, steps), run<MODE, 2>(hbm, hbmmask, out, steps),
           run<MODE, 3>(hbm, hbmmask, out, steps));
  printf("   TFLOP/s (peak 157.3)\n");
  fflush(stdout);
}

int main() {
  const size_t l2_bytes = 2u << 20, hbm_bytes = 1u << 30;     // 2 MiB (L2-resident per XCD), 1 GiB
  float *l2, *hbm, *out;
  hipMalloc(&l2, l2_bytes + (1 << 20)); hipMalloc(&hbm, hbm_bytes + (1 << 20)); hipMalloc(&out, 768 * 256 * 4);
  hipMemset(l2, 0, l2_bytes + (1 << 20)); hipMemset(hbm, 0, hbm_bytes + (1 << 20));
  const unsigned l2mask = (unsigned)l2_bytes - 1u, hbmmask = (unsigned)hbm_bytes - 1u;
  const int steps = 1500;
  row<0>("MFMA chain only", l2, l2mask, hbm, hbmmask, out, steps);
  row<BAR>("+ barrier", l2, l2mask, hbm, hbmmask, out, steps);
  row<BAR | FR>("+ fragment reads", l2, l2mask, hbm, hbmmask, out, steps);
  row<BAR | FR | ST>("+ ds_write_b128 staging", l2, l2mask, hbm, hbmmask, out, steps);
  row<BAR | FR | LD>("+ buffer loads, no LDS stores", l2, l2mask, hbm, hbmmask, out, steps);
  row<BAR | FR | ST | LD>("+ loads + stores (product kernel)", l2, l2mask, hbm, hbmmask, out, steps);
  row<BAR | FR | DMA>("+ buffer_load ... lds (LDS-DMA)", l2, l2mask, hbm, hbmmask, out, steps);
  row<FR | ST | LD>("loads + stores, NO barrier", l2, l2mask, hbm, hbmmask, out, steps);
  row<BAR | FR4>("4 fragment sets (12 MFMAs ahead)", l2, l2mask, hbm, hbmmask, out, steps);
  row<BAR | FR4 | ST | LD>("4 fragment sets + loads + stores", l2, l2mask, hbm, hbmmask, out, steps);
  row<BAR | FR | AG>("fragment reads, acc in AGPRs", l2, l2mask, hbm, hbmmask, out, steps);
  row<BAR | FR | ST | LD | AG>("product step, acc in AGPRs", l2, l2mask, hbm, hbmmask, out, steps);
  // the product kernel's operand rows are not contiguous: 128-byte row pieces at the activation / weight row strides
  for (unsigned stride : {1552u, 4608u}) {
    g_row_stride = stride;
    char name[64];
    snprintf(name, sizeof name, "product step, rows %u B apart", stride);
    row<BAR | FR | ST | LD>(name, l2, l2mask, hbm, hbmmask, out, steps);
  }
  g_row_stride = 128;
  run_bf16<1>(out, 0.0); run_bf16<2>(out, 0.0); run_bf16<2>(out, 1.5); run_bf16<1>(out, 1.5);
  // sustained shader clock: the same bodies after 1.5 s of back-to-back launches; MHz = s_memtime / s_memrealtime * 100
  g_sustain_s = 1.5;
  printf("sustained (1.5 s of launches first), x3 blocks per CU:\n");
  { double tf = run<0, 3>(l2, l2mask, out, steps);                  printf("  MFMA chain only        %6.1f TFLOP/s  %7.1f MHz\n", tf, g_last_mhz); }
  { double tf = run<BAR, 3>(l2, l2mask, out, steps);                printf("  + barrier              %6.1f TFLOP/s  %7.1f MHz\n", tf, g_last_mhz); }
  { double tf = run<BAR | FR, 3>(l2, l2mask, out, steps);           printf("  + fragment reads       %6.1f TFLOP/s  %7.1f MHz\n", tf, g_last_mhz); }
  { double tf = run<BAR | FR | ST | LD, 3>(l2, l2mask, out, steps); printf("  product step (L2)      %6.1f TFLOP/s  %7.1f MHz\n", tf, g_last_mhz); }
  { double tf = run<BAR | FR | ST | LD, 3>(hbm, hbmmask, out, steps); printf("  product step (HBM)     %6.1f TFLOP/s  %7.1f MHz\n", tf, g_last_mhz); }
  g_sustain_s = 0.0;
  { double tf = run<BAR | FR | ST | LD, 3>(l2, l2mask, out, steps); printf("  product step, cold     %6.1f TFLOP/s  %7.1f MHz\n", tf, g_last_mhz); }
  return 0;
}
