// The student's CLASSIFIER as one row-local launch (round 6): for a large batch in front of a narrow last layer (<= 48 classes)
//   logits = tail(z) . W^T + b   ->   log_softmax   ->   NLL / KL(log-target)   ->   dlogits
// (reference models.py:45-52 last layer + train_and_eval.py:77-84 with the criteria of train_student.py:278-279) by workgroups that own
// 16 rows each.  It replaces a split-K GEMM launch over 64 x 64 tiles (eight K slabs: 17.9 us for MLP3w8's 32 MB of z) and the loss
// launch behind it (5.8 us).  The sixteen waves of a workgroup split the hidden width; a wave streams its [16 rows] x [K / 16] piece of z
// ONCE, straight into the A operand of v_mfma_f32_16x16x4_f32 (a float4 of a row = the k values of four consecutive MFMAs: any pairing of
// k is a valid summation order) with the hidden layer's tail -- BatchNorm affine, ReLU, counter-based dropout: the expressions of gemm.hip's
// operand transform XF == 2 -- evaluated in registers; the W fragments (47 x K floats: L2 hits) are float4 loads of the same shape.  The
// sixteen 16 x 48 partial tiles are summed through LDS in wave order (fixed: run-to-run deterministic), the bias is added, the logits are
// stored, and the rows' softmax / loss / gradient follow with ONE WAVE PER ROW in the arithmetic of softmax_loss_kernel<true, 64>: same
// expressions, same shuffle trees, and per-four-rows loss partials laid out like that kernel's 1024-workgroup form -- so the loss and
// dlogits are the bits the two-launch form produces from the same logits (tests/test_kernels_gpu.py).
#include <type_traits>
#include <utility>

#include "glnn_common.h"
#include "student_dev.h"

namespace {

typedef float cls_f32x4 __attribute__((ext_vector_type(4)));

struct ClsArgs {
  const float* a; int64_t lda; int64_t m; int k;                 // the hidden rows: z (XF) or the stored tail (plain)
  const float* a_scale; const float* a_shift; uint32_t dthr; uint32_t dseed; float dscale;
  const float* w; int64_t ldw; int n; const float* bias;         // W [n][k], n <= 48
  float* logits; int64_t ldz;
  int with_loss;
  LossArgs loss;                                                 // rows / c / kind / labels / targets / scale / dz / ldg / logp / partial
};

constexpr int kClsRows = 16;          // rows per workgroup = M of the MFMA
constexpr int kClsWaves = 16;         // waves per workgroup = the K split (four per SIMD: the loads of one hide behind the MFMAs of the others)
constexpr int kClsLd = 49;            // LDS pitch of a 16 x 48 tile (odd: the column-wise writes of a C fragment spread over the banks)
constexpr int kClsMaxK = 4096;        // hidden width the constants' LDS copy holds

__device__ __forceinline__ float4 cls_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void cls_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ float cls_bperm(int addr, float x) { return __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(x))); }
__device__ __forceinline__ float4 cls_perm(int addr, const float4& x) {
  return make_float4(cls_bperm(addr, x.x), cls_bperm(addr, x.y), cls_bperm(addr, x.z), cls_bperm(addr, x.w));
}
__device__ __forceinline__ float4 cls_perm(int addr, const cls_f32x4& x) {
  return make_float4(cls_bperm(addr, x[0]), cls_bperm(addr, x[1]), cls_bperm(addr, x[2]), cls_bperm(addr, x[3]));
}

template <class F, int... I>
__device__ __forceinline__ void cls_static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void cls_static_for(F&& f) { cls_static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// one 16-column step of a wave: tail of the four z values of the lane, then the twelve MFMAs (three class blocks x four k)
template <int XF>
__device__ __forceinline__ void cls_step(const ClsArgs& g, const float* cst, int c, uint32_t hrow, float4 xx, const float4& b0, const float4& b1,
                                         const float4& b2, cls_f32x4 (&acc)[3]) {
  if (XF) {
    const float4 sc = *reinterpret_cast<const float4*>(cst + c), sh = *reinterpret_cast<const float4*>(cst + g.k + c);
    xx.x = fmaxf(fmaf(xx.x, sc.x, sh.x), 0.f);
    xx.y = fmaxf(fmaf(xx.y, sc.y, sh.y), 0.f);
    xx.z = fmaxf(fmaf(xx.z, sc.z, sh.z), 0.f);
    xx.w = fmaxf(fmaf(xx.w, sc.w, sh.w), 0.f);
    if (XF == 2) {
      // glnn::drop_keep of columns c .. c + 3 (c % 4 == 0): one hash per pair of adjacent columns, 16 bits of it per element
      const uint32_t c2 = (uint32_t)c >> 1;
      uint32_t h0 = hrow ^ (c2 * 0x85EBCA77u + 0x632BE5ABu), h1 = hrow ^ ((c2 + 1u) * 0x85EBCA77u + 0x632BE5ABu);
      h0 ^= h0 >> 16; h0 *= 0x7feb352du; h0 ^= h0 >> 15; h0 *= 0x846ca68bu; h0 ^= h0 >> 16;
      h1 ^= h1 >> 16; h1 *= 0x7feb352du; h1 ^= h1 >> 15; h1 *= 0x846ca68bu; h1 ^= h1 >> 16;
      xx.x = (h0 & 0xFFFFu) >= g.dthr ? xx.x * g.dscale : 0.f;
      xx.y = (h0 >> 16) >= g.dthr ? xx.y * g.dscale : 0.f;
      xx.z = (h1 & 0xFFFFu) >= g.dthr ? xx.z * g.dscale : 0.f;
      xx.w = (h1 >> 16) >= g.dthr ? xx.w * g.dscale : 0.f;
    }
  }
  acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(xx.x, b0.x, acc[0], 0, 0, 0);
  acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(xx.x, b1.x, acc[1], 0, 0, 0);
  acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(xx.x, b2.x, acc[2], 0, 0, 0);
  acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(xx.y, b0.y, acc[0], 0, 0, 0);
  acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(xx.y, b1.y, acc[1], 0, 0, 0);
  acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(xx.y, b2.y, acc[2], 0, 0, 0);
  acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(xx.z, b0.z, acc[0], 0, 0, 0);
  acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(xx.z, b1.z, acc[1], 0, 0, 0);
  acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(xx.z, b2.z, acc[2], 0, 0, 0);
  acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(xx.w, b0.w, acc[0], 0, 0, 0);
  acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(xx.w, b1.w, acc[1], 0, 0, 0);
  acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(xx.w, b2.w, acc[2], 0, 0, 0);
}

// XF: 0 plain operand, 1 affine + ReLU, 2 affine + ReLU + dropout.  STEPS > 0: the wave's k / 256 sixteen-column steps, fully unrolled,
// the loads of step j + DEPTH issued before step j is used (a rolling window: the counter waits of the unrolled code never drain the
// queue); STEPS == 0: any step count, one step per trip.
#ifndef GLNN_CLS_DEPTH
#define GLNN_CLS_DEPTH 3
#endif
constexpr int kClsDepth = GLNN_CLS_DEPTH;   // steps of loads in flight ahead of the step in the matrix cores
template <int XF, int STEPS>
__global__ __launch_bounds__(1024) void cls_fwd_kernel(const ClsArgs g) {
  constexpr int UNR = 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int64_t r0 = (int64_t)blockIdx.x * kClsRows;
  const int64_t row = r0 + r < g.m ? r0 + r : g.m - 1;           // rows past the batch: the last row's values, never stored
  const int kw = g.k / kClsWaves;                                // columns per wave: a multiple of 16
  const int kb = wave * kw + 4 * q;
  // LOADS: lane = 4 * row + segment -- the four lanes of a quad read 64 contiguous bytes of ONE row, which the texture addresser merges into one
  // request.  In the MFMA's own operand order (lane = row + 16 * segment) the lanes of a quad sit on four different rows = four requests of
  // 16 bytes each: 64 requests per load instruction instead of 16, and the launch was bound by exactly that (17.7 us whatever the prefetch
  // depth and the workgroup count).  The registers are brought into operand order by ds_bpermute_b32 (the LDS crossbar, no LDS memory).
  const int lr = lane >> 2, lsg = lane & 3;
  const int64_t ldrow = r0 + lr < g.m ? r0 + lr : g.m - 1;
  const float* ap = g.a + ldrow * g.lda + wave * kw + 4 * lsg;
  const float* wp[3];
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const int cls = 16 * t + lr < g.n ? 16 * t + lr : g.n - 1;   // classes past n: the last class's weights into columns that are never used
    wp[t] = g.w + (int64_t)cls * g.ldw + wave * kw + 4 * lsg;
  }
  const int paddr = 4 * (4 * r + q);                             // operand lane (r, q) takes the register of load lane 4 r + q
  // per-column constants of the tail: one LDS copy per workgroup ([0, k) scale, [k, 2 k) shift) -- as global loads they were a third of
  // the kernel's vector-memory instructions, every one of them the same 64 bytes for sixteen lanes
  extern __shared__ __attribute__((aligned(16))) float cst[];
  if (XF) {
    for (int e = 4 * tid; e < 2 * g.k; e += 4 * 1024)
      *reinterpret_cast<float4*>(cst + e) = e < g.k ? cls_ld4(g.a_scale + e) : cls_ld4(g.a_shift + (e - g.k));
  }
  // the criterion's operands of the row this wave will finish (row r0 + wave): requested now, used behind the product
  const int64_t lrow = r0 + wave;
  float tj0 = 0.f;
  int64_t yrow = 0;
  if (g.with_loss && lrow < g.m) {
    const LossArgs& a = g.loss;
    if (a.kind == GLNN_LOSS_NLL) yrow = a.labels[a.label_rows ? a.label_rows[lrow] : lrow];
    else tj0 = lane < a.c ? a.t[(a.t_rows ? a.t_rows[lrow] : lrow) * a.ldt + lane] : 0.f;
  }
  cls_f32x4 acc[3];
#pragma unroll
  for (int t = 0; t < 3; ++t) acc[t] = cls_f32x4{0.f, 0.f, 0.f, 0.f};
  const uint32_t hrow = g.dseed ^ ((uint32_t)row * 0x9E3779B1u);
  if constexpr (STEPS > 0) {
    // The rolling window is stated in asm: left to the compiler, every load sinks down to its first use (its scheduler minimises register
    // pressure) and the wave drains its queue once per step -- the launch then takes four memory round trips per workgroup whatever the
    // chip has free (measured: 21 us, the same for 128 and 256 workgroups).  Loads return in order: behind `s_waitcnt vmcnt(N)` with
    // N = the loads issued after step j's, step j's four registers are valid; the "+v" operands tie the step's uses to that wait.
    cls_f32x4 v[STEPS], w0[STEPS], w1[STEPS], w2[STEPS];
    constexpr int D = kClsDepth < STEPS ? kClsDepth : STEPS;
    auto issue = [&](auto j_) {
      constexpr int j = decltype(j_)::value;
      (void)v; (void)w0; (void)w1; (void)w2; (void)ap; (void)wp;
      asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(v[j]) : "v"(ap), "n"(64 * j) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(w0[j]) : "v"(wp[0]), "n"(64 * j) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(w1[j]) : "v"(wp[1]), "n"(64 * j) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(w2[j]) : "v"(wp[2]), "n"(64 * j) : "memory");
    };
    cls_static_for<D>(issue);
    if (XF) __syncthreads();
    cls_static_for<STEPS>([&](auto j_) {
      constexpr int j = decltype(j_)::value;
      (void)v; (void)w0; (void)w1; (void)w2; (void)acc; (void)cst; (void)hrow; (void)kb;
      if constexpr (j + D < STEPS) issue(std::integral_constant<int, j + D>{});
      constexpr int newer = 4 * ((STEPS - 1 - j) < D ? (STEPS - 1 - j) : D);
      asm volatile("s_waitcnt vmcnt(%4)" : "+v"(v[j]), "+v"(w0[j]), "+v"(w1[j]), "+v"(w2[j]) : "n"(newer) : "memory");
      cls_step<XF>(g, cst, kb + 16 * j, hrow, cls_perm(paddr, v[j]), cls_perm(paddr, w0[j]), cls_perm(paddr, w1[j]), cls_perm(paddr, w2[j]), acc);
    });
  } else {
  const int steps = kw / 16;
  float4 v[UNR], w0[UNR], w1[UNR], w2[UNR];
#pragma unroll
  for (int u = 0; u < UNR; ++u) {                                // the first group is in flight while the constants land
    v[u] = cls_ld4(ap + 16 * u);
    w0[u] = cls_ld4(wp[0] + 16 * u); w1[u] = cls_ld4(wp[1] + 16 * u); w2[u] = cls_ld4(wp[2] + 16 * u);
  }
  if (XF) __syncthreads();
  for (int j0 = 0; j0 < steps; j0 += UNR) {
    float4 x[UNR], b0[UNR], b1[UNR], b2[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) { x[u] = v[u]; b0[u] = w0[u]; b1[u] = w1[u]; b2[u] = w2[u]; }
    if (j0 + UNR < steps) {
#pragma unroll
      for (int u = 0; u < UNR; ++u) {                            // the next group's loads are issued before this group is used
        const int o = 16 * (j0 + UNR + u);
        v[u] = cls_ld4(ap + o);
        w0[u] = cls_ld4(wp[0] + o); w1[u] = cls_ld4(wp[1] + o); w2[u] = cls_ld4(wp[2] + o);
      }
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u)
      cls_step<XF>(g, cst, kb + 16 * (j0 + u), hrow, cls_perm(paddr, x[u]), cls_perm(paddr, b0[u]), cls_perm(paddr, b1[u]), cls_perm(paddr, b2[u]), acc);
  }
  }
  // C fragment of v_mfma_f32_16x16x4_f32: register v of lane (r, q) is row 4 q + v, column r of the 16 x 16 block
  __shared__ float red[kClsWaves][kClsRows][kClsLd];
  __shared__ float lg[kClsRows][kClsLd];
  __shared__ float wl[kClsWaves];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) red[wave][4 * q + i][16 * t + r] = acc[t][i];
  // Barriers behind which only LDS traffic matters are stated as such (wait for this wave's LDS operations, then s_barrier):
  // __syncthreads() also waits for every global STORE of the wave to be acknowledged -- with the logits / dlogits stores in front of them
  // the two barriers of the loss part cost the launch 4.5 us.  The logits are stored last, from the register that summed them.
  cls_lds_barrier();
  float logit = 0.f;
  const int lrr = tid / 48, lcc = tid - lrr * 48;
  const bool lvalid = tid < kClsRows * 48 && lcc < g.n;
  if (tid < kClsRows * 48) {
    float s = 0.f;
#pragma unroll
    for (int wv = 0; wv < kClsWaves; wv += 4)                    // wave order, in fours: fixed
      s += (red[wv][lrr][lcc] + red[wv + 1][lrr][lcc]) + (red[wv + 2][lrr][lcc] + red[wv + 3][lrr][lcc]);
    if (lcc < g.n) {
      s += g.bias ? g.bias[lcc] : 0.f;
      lg[lrr][lcc] = s;
      logit = s;
    }
  }
  if (!g.with_loss) {
    if (lvalid && r0 + lrr < g.m) g.logits[(r0 + lrr) * g.ldz + lcc] = logit;
    return;
  }
  cls_lds_barrier();
  // ---- one wave per row: softmax_loss_kernel<true, 64>'s row, c <= 64 (one class per lane) ----
  const LossArgs& a = g.loss;
  {
    const int rr = wave;
    const int64_t rw = r0 + rr;
    float row_loss = 0.f;
    if (rw < g.m) {
      const float zl = lane < a.c ? lg[rr][lane] : 0.f;
      float mx = -INFINITY;
      if (lane < a.c) mx = fmaxf(mx, zl);
      mx = wave_max(mx);
      float se = 0.f;
      if (lane < a.c) se += expf(zl - mx);
      se = wave_sum(se);
      const float lse = mx + logf(se);
      if (a.kind == GLNN_LOSS_NLL) {
        const int64_t y = yrow;
        if (lane < a.c) {
          const int j = lane;
          const float lp = zl - lse;
          if (a.logp) a.logp[rw * a.ldl + j] = lp;
          const float sm = expf(lp);
          const float gd = (sm - (j == y ? 1.f : 0.f)) * a.scale;
          a.dz[rw * a.ldg + j] = gd;
          if (j == y) row_loss = -lp;
        }
        row_loss = wave_sum(row_loss);
      } else {
        float set = 0.f;
        if (lane < a.c) {
          const float tj = tj0, et = expf(tj);
          set += et;
          row_loss += et * (tj - (zl - lse));
        }
        set = wave_sum(set);
        row_loss = wave_sum(row_loss);
        if (lane < a.c) {
          const int j = lane;
          const float lp = zl - lse;
          if (a.logp) a.logp[rw * a.ldl + j] = lp;
          const float gd = (expf(lp) * set - expf(tj0)) * a.scale;
          a.dz[rw * a.ldg + j] = gd;
        }
      }
    }
    if (lane == 0) wl[wave] = row_loss;
    cls_lds_barrier();
    // the loss kernel's workgroup b owns rows 4 b .. 4 b + 3 and stores (s0 + s1) + (s2 + s3): four such groups per workgroup here
    if (tid < 4) {
      const int64_t b = r0 / 4 + tid;
      if (4 * b < g.m) a.partial[b] = (wl[4 * tid] + wl[4 * tid + 1]) + (wl[4 * tid + 2] + wl[4 * tid + 3]);
    }
  }
  if (lvalid && r0 + lrr < g.m) g.logits[(r0 + lrr) * g.ldz + lcc] = logit;
}

}  // namespace

// The classifier product of a large batch (+ loss) in one launch; GLNN_ERR_UNSUPPORTED = nothing launched (any other shape: the GEMM
// and loss launches).  ls == NULL: logits only.  ls->pf / the finalize launch as in glnn::softmax_loss's counter-less form.
int glnn::cls_fwd(const float* a, int64_t lda, const float* a_scale, const float* a_shift, float drop_p, uint32_t drop_seed, int64_t m, int k,
                  const float* w, int64_t ldw, int n, const float* bias, float* logits, int64_t ldz, const ClsLoss* ls, void* stream) {
  if (!glnn::opts().cls_fused) return GLNN_ERR_UNSUPPORTED;
  if (m <= 1024 || n < 1 || n > 48 || k < 256 || k % 256 != 0 || k > kClsMaxK || lda % 4 != 0 || ldw % 4 != 0 || lda < k || ldw < k || ldz < n)
    return GLNN_ERR_UNSUPPORTED;
  if (!glnn::aligned16(a) || !glnn::aligned16(w) || (a_scale && (!a_shift || !glnn::aligned16(a_scale) || !glnn::aligned16(a_shift))))
    return GLNN_ERR_UNSUPPORTED;
  if (!a_scale && drop_p > 0.f) return GLNN_ERR_UNSUPPORTED;
  if ((m + kClsRows - 1) / kClsRows > 0x7FFFFFFF) return GLNN_ERR_UNSUPPORTED;
  ClsArgs g = {};
  g.a = a; g.lda = lda; g.m = m; g.k = k; g.a_scale = a_scale; g.a_shift = a_shift;
  g.dthr = drop_p > 0.f ? glnn::drop_threshold(drop_p) : 0u; g.dseed = drop_seed; g.dscale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.f;
  g.w = w; g.ldw = ldw; g.n = n; g.bias = bias; g.logits = logits; g.ldz = ldz;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  int64_t blocks = 0;
  if (ls) {
    blocks = (m + 3) / 4;
    if (blocks > 1024 || ls->ws_floats < blocks || !ls->ws || !ls->dlogits || ls->ldg < n) return GLNN_ERR_UNSUPPORTED;
    if (ls->kind == GLNN_LOSS_NLL && !ls->labels) return GLNN_ERR_UNSUPPORTED;
    if (ls->kind == GLNN_LOSS_KL && (!ls->target_logp || ls->ldt < n)) return GLNN_ERR_UNSUPPORTED;
    LossArgs& l = g.loss;
    l.z = logits; l.ldz = ldz; l.rows = m; l.c = n; l.kind = ls->kind; l.labels = ls->labels; l.label_rows = ls->label_rows;
    l.t = ls->target_logp; l.ldt = ls->ldt; l.t_rows = ls->target_rows; l.scale = ls->lamb / (float)m;
    l.dz = ls->dlogits; l.ldg = ls->ldg; l.logp = nullptr; l.ldl = 0; l.partial = ls->ws; l.inv_rows = 1.0f / (float)m;
    g.with_loss = 1;
  }
  const dim3 grid((unsigned)((m + kClsRows - 1) / kClsRows));
  const int xf = !a_scale ? 0 : (g.dthr ? 2 : 1);
  const int steps = k / (kClsWaves * 16);
  const size_t lds = xf ? (size_t)2 * k * sizeof(float) : 0;
  const dim3 blk(kClsWaves * 64);
#define GLNN_CLS_LAUNCH(XF_, S_) hipLaunchKernelGGL((cls_fwd_kernel<XF_, S_>), grid, blk, lds, st, g)
#define GLNN_CLS_STEPS(XF_) do { if (steps == 8) GLNN_CLS_LAUNCH(XF_, 8); else if (steps == 4) GLNN_CLS_LAUNCH(XF_, 4); else if (steps == 2) GLNN_CLS_LAUNCH(XF_, 2); \
                                 else if (steps == 1) GLNN_CLS_LAUNCH(XF_, 1); else if (steps == 16) GLNN_CLS_LAUNCH(XF_, 16); else GLNN_CLS_LAUNCH(XF_, 0); } while (0)
  if (xf == 0) GLNN_CLS_STEPS(0); else if (xf == 1) GLNN_CLS_STEPS(1); else GLNN_CLS_STEPS(2);
#undef GLNN_CLS_STEPS
#undef GLNN_CLS_LAUNCH
  const int rc = glnn::check_launch("glnn::cls_fwd");
  if (rc != GLNN_OK) return rc;
  if (ls) {
    if (ls->pf && !ls->pf->has_loss) {
      ls->pf->has_loss = 1;
      ls->pf->loss = {ls->ws, (int)blocks, 1.0f / (float)m, ls->loss_out, ls->loss_accum};
    } else {
      return glnn::loss_finalize(ls->ws, (int)blocks, 1.0f / (float)m, ls->loss_out, ls->loss_accum, stream);
    }
  }
  return GLNN_OK;
}

extern "C" int glnn_classifier_loss_f32(const float* a, int64_t lda, const float* a_scale, const float* a_shift, float drop_p,
                                        uint32_t drop_seed, int64_t rows, int k, const float* w, int64_t ldw, int c, const float* bias,
                                        float* logits, int64_t ldz, int kind, const int64_t* labels, const int64_t* label_rows,
                                        const float* target_logp, int64_t ldt, const int64_t* target_rows, float lamb, float* dlogits,
                                        int64_t ldg, float* loss_out, float* loss_accum, float* workspace, int64_t workspace_floats,
                                        void* stream) {
  GLNN_REQUIRE(a && w && logits, "glnn_classifier_loss_f32: null pointer");
  GLNN_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "glnn_classifier_loss_f32: drop_p outside [0, 1)");
  if (kind < 0) return glnn::cls_fwd(a, lda, a_scale, a_shift, drop_p, drop_seed, rows, k, w, ldw, c, bias, logits, ldz, nullptr, stream);
  GLNN_REQUIRE(kind == GLNN_LOSS_NLL || kind == GLNN_LOSS_KL, "glnn_classifier_loss_f32: unknown kind %d", kind);
  GLNN_REQUIRE(dlogits && workspace && loss_out, "glnn_classifier_loss_f32: a criterion needs dlogits, loss_out and the workspace");
  const glnn::ClsLoss cl = {kind, labels, label_rows, target_logp, ldt, target_rows, lamb, dlogits, ldg, loss_out, loss_accum, workspace,
                            workspace_floats, nullptr};
  return glnn::cls_fwd(a, lda, a_scale, a_shift, drop_p, drop_seed, rows, k, w, ldw, c, bias, logits, ldz, &cl, stream);
}
