// Micro-benchmark of the per-symbol loop of k_symbols_reg: where do the cycles of one iteration go?
// hipcc --offload-arch=gfx950 -O3 -o rans_loop rans_loop.hip && ./rans_loop
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef uint32_t v32u __attribute__((ext_vector_type(32)));

#define BODY_COMMON_PRE \
      "Lt%=:\n" \
      " s_cmpk_lt_u32 s21, 0x4000\n" \
      " s_cbranch_scc1 Lren%=\n" \
      " v_cmp_eq_u32_e32 vcc, %[r], %[key]\n" \
      " s_bfe_u32 %[k6], s21, 0x60006\n" \
      " v_mov_b32_e32 %[va], s21\n" \
      " s_lshr_b32 %[q], s21, 12\n" \
      " v_cndmask_b32_e32 %[mine], %[mine], %[va], vcc\n"
#define BODY_COMMON_POST \
      " v_lshrrev_b32_e32 %[vf], 12, %[va]\n" \
      " v_and_b32_e32 %[va], 0xfff, %[va]\n" \
      " v_mad_u32_u24 %[va], %[vf], %[q], %[va]\n" \
      " s_sub_u32 %[r], %[r], 1\n"
#define TAIL \
      " s_cbranch_scc0 Lt%=\n" \
      " s_branch Lend%=\n" \
      "Lren%=:\n" \
      " s_sub_u32 %[rc], %[rc], 1\n" \
      " s_cbranch_scc1 Lempty%=\n" \
      " s_lshl_b64 s[20:21], s[20:21], 8\n" \
      " s_branch Lt%=\n" \
      "Lempty%=:\n" \
      " s_mov_b32 %[rc], 0\n" \
      "Lend%=:\n"
#define OPS : "+{s[20:21]}"(P), [rc] "+s"(rc), [r] "+s"(r), [mine] "+v"(mine), [k6] "=&s"(k6), [q] "=&s"(q), [va] "=&v"(va), [vf] "=&v"(vf) \
            : [key] "v"(key), "{v[16:47]}"(lo), "{v[48:79]}"(hi) : "vcc", "scc"

template <int VARIANT>
__global__ __launch_bounds__(64) void k(const uint32_t *tab, uint32_t *out, uint32_t iters, uint64_t *cycles) {
  v32u lo, hi;
  const uint32_t lane = threadIdx.x;
#pragma unroll
  for (int i = 0; i < 32; ++i) { lo[i] = tab[i * 64 + lane]; hi[i] = tab[(i + 32) * 64 + lane]; }
  uint64_t P = ((uint64_t)0x12345u << 32) | 0xFFFFFFFFu;
  uint32_t rc = 0x7FFFFFFF, mine = 0, k6, q, va, vf;
  const uint32_t key = 63 - lane;
  uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (uint32_t it = 0; it < iters; ++it) {
    uint32_t r = 63;
    if (VARIANT == 0) {             // the loop of k_symbols_reg
      asm volatile(BODY_COMMON_PRE
        " s_set_gpr_idx_on %[k6], gpr_idx(SRC0)\n v_mov_b32_e32 %[va], v16\n s_set_gpr_idx_off\n"
        BODY_COMMON_POST " v_readlane_b32 s21, %[va], s21\n" TAIL OPS);
    } else if (VARIANT == 1) {      // without the capture of x into lane j (v_cmp, v_mov, v_cndmask)
      asm volatile("Lt%=:\n s_cmpk_lt_u32 s21, 0x4000\n s_cbranch_scc1 Lren%=\n s_bfe_u32 %[k6], s21, 0x60006\n s_lshr_b32 %[q], s21, 12\n"
        " s_set_gpr_idx_on %[k6], gpr_idx(SRC0)\n v_mov_b32_e32 %[va], v16\n s_set_gpr_idx_off\n"
        BODY_COMMON_POST " v_readlane_b32 s21, %[va], s21\n" TAIL OPS);
    } else if (VARIANT == 2) {      // ... and without the row arithmetic: readlane straight from the selected row
      asm volatile("Lt%=:\n s_cmpk_lt_u32 s21, 0x4000\n s_cbranch_scc1 Lren%=\n s_bfe_u32 %[k6], s21, 0x60006\n s_lshr_b32 %[q], s21, 12\n"
        " s_set_gpr_idx_on %[k6], gpr_idx(SRC0)\n v_mov_b32_e32 %[va], v16\n s_set_gpr_idx_off\n"
        " s_sub_u32 %[r], %[r], 1\n v_readlane_b32 s21, %[va], s21\n" TAIL OPS);
    } else if (VARIANT == 3) {      // ... and without the window: a single v_readlane per symbol is the only VALU instruction
      asm volatile("Lt%=:\n s_cmpk_lt_u32 s21, 0x4000\n s_cbranch_scc1 Lren%=\n s_bfe_u32 %[k6], s21, 0x60006\n s_lshr_b32 %[q], s21, 12\n"
        " s_sub_u32 %[r], %[r], 1\n v_readlane_b32 s21, v16, s21\n" TAIL OPS);
    } else if (VARIANT == 4) {      // pure scalar chain
      asm volatile("Lt%=:\n s_cmpk_lt_u32 s21, 0x4000\n s_cbranch_scc1 Lren%=\n s_bfe_u32 %[k6], s21, 0x60006\n s_lshr_b32 %[q], s21, 12\n"
        " s_mul_i32 s21, s21, 0x19660d\n s_bfe_u32 s21, s21, 0x160008\n s_or_b32 s21, s21, 0x4000\n s_sub_u32 %[r], %[r], 1\n" TAIL OPS);
    }
  }
  uint64_t t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) { cycles[blockIdx.x] = t1 - t0; }
  out[blockIdx.x * 64 + lane] = mine + (uint32_t)(P >> 32) + rc;
}

int main() {
  uint32_t *tab, *out; uint64_t *cyc;
  hipMalloc(&tab, 4096 * 4); hipMalloc(&out, 8192 * 64 * 4); hipMalloc(&cyc, 8192 * 8);
  uint32_t h[4096];
  for (int i = 0; i < 4096; ++i) h[i] = 0x4000u | (uint32_t)((i * 2654435761u) >> 12 & 0x3FFFFF);   // rows hold next states >= 0x4000 (variants 2,3); variants 0,1: f = e>>12 >= 4
  hipMemcpy(tab, h, sizeof(h), hipMemcpyHostToDevice);
  const uint32_t iters = 2000;
  for (int waves : {1, 256 * 4, 256 * 20}) {
    uint64_t hc[5] = {0, 0, 0, 0, 0};
    for (int v = 0; v < 5; ++v) {
      for (int rep = 0; rep < 2; ++rep) {
        if (v == 0) hipLaunchKernelGGL(k<0>, dim3(waves), dim3(64), 0, 0, tab, out, iters, cyc);
        if (v == 1) hipLaunchKernelGGL(k<1>, dim3(waves), dim3(64), 0, 0, tab, out, iters, cyc);
        if (v == 2) hipLaunchKernelGGL(k<2>, dim3(waves), dim3(64), 0, 0, tab, out, iters, cyc);
        if (v == 3) hipLaunchKernelGGL(k<3>, dim3(waves), dim3(64), 0, 0, tab, out, iters, cyc);
        if (v == 4) hipLaunchKernelGGL(k<4>, dim3(waves), dim3(64), 0, 0, tab, out, iters, cyc);
        hipDeviceSynchronize();
      }
      uint64_t c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
      hc[v] = c;
    }
    printf("waves %5d: cycles/symbol  full %.1f | no capture %.1f | no capture, no row math %.1f | readlane only %.1f | scalar only %.1f\n", waves,
           hc[0] / (64.0 * iters), hc[1] / (64.0 * iters), hc[2] / (64.0 * iters), hc[3] / (64.0 * iters), hc[4] / (64.0 * iters));
  }
  return 0;
}
