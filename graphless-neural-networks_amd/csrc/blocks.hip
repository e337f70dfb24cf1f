// Block construction and CSR transposition on the GPU: the integer / index half of the teacher TRAINING path
// (SURVEY.md section 8f rows 1 and 2).  HBM- and latency-bound integer work, sized by the batch's FRONTIER, never by N.
//
// Replaces (see include/glnn_hip.h):
//   * dgl.dataloading.NodeDataLoader's per-batch block construction -- `to_block` relabelling of the sampled /
//     full-neighbour frontier (reference train_and_eval.py:176-205, consumed at :41 and models.py:109,134-137):
//     glnn_block_build = row counts -> exclusive scan (block indptr) -> hash-table insert of the frontier's node ids
//     -> scan over first occurrences (dense local ids, "destinations first, then first appearance") -> relabel;
//   * the reversed graph autograd walks in loss.backward() through dgl's SpMM (train_and_eval.py:27,54):
//     glnn_csr_transpose = counting sort by source (histogram -> scan -> atomic fill) + a canonical per-row sort, so
//     that A^T dY runs on the SAME gather kernel (glnn_spmm_csr_f32) deterministically.
//
// Building blocks
//   * scan_kernel: single-pass exclusive scan with decoupled look-back (one launch; workgroup order taken from an
//     atomic ticket so that every predecessor a workgroup waits on is already resident); value source and consumer are
//     functors, so the producer (row counts, first-occurrence flags, histogram) and the consumer (indptr write, id
//     assignment, cursor reset) are fused into the scan instead of being separate passes;
//   * an open-addressing hash table (linear probing, atomicCAS on int32 keys) over 2..4x the frontier size.
// Workspaces arrive filled with the byte 0x7F by ONE hipMemsetAsync: 0x7F7F7F7F is the table's EMPTY key / "no position"
// value, the look-back status words read as state EMPTY, and the ticket counters start at 0x7F7F7F7F.
#include "glnn_common.h"

namespace {

constexpr int kFill32 = 0x7F7F7F7F;
constexpr int kScanThreads = 256;
constexpr int kScanItems = 4;
constexpr int kScanTile = kScanThreads * kScanItems;

// look-back status word: top 2 bits = state (01 = EMPTY, what the 0x7F memset leaves; 10 = aggregate; 11 = inclusive prefix)
constexpr unsigned long long kStAgg = 2ull << 62, kStPrefix = 3ull << 62, kStMask = (1ull << 62) - 1;

// The status word carries its own payload (state + value in ONE 64-bit word), so nothing else has to become visible with
// it: relaxed agent-scope atomics (write-through / L2-bypassing accesses) are enough.  Release / acquire at agent scope
// would write back / invalidate the XCD's whole L2 in every workgroup -- on this 8-XCD part that is what a device-scope
// fence costs.
__device__ __forceinline__ void st_release(unsigned long long* p, unsigned long long v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long ld_acquire(unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// out(i, exclusive prefix, value) for i in [0,n); *total = sum.  status: >= ceil(n/1024) words, ticket: 1 int, both 0x7F-filled.
template <class V, class W>
__global__ __launch_bounds__(kScanThreads) void scan_kernel(V val, int64_t n, unsigned long long* status, int* ticket, W out,
                                                            int64_t* total) {
  __shared__ int s_bid;
  __shared__ int s_wave[kScanThreads / 64];
  __shared__ long long s_prefix;
  if (threadIdx.x == 0) s_bid = atomicAdd(ticket, 1) - kFill32;
  __syncthreads();
  const int bid = s_bid;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t base = (int64_t)bid * kScanTile + (int64_t)threadIdx.x * kScanItems;
  int v[kScanItems];
  int tsum = 0;
#pragma unroll
  for (int t = 0; t < kScanItems; ++t) {
    v[t] = (base + t < n) ? val(base + t) : 0;
    tsum += v[t];
  }
  int inc = tsum;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int y = __shfl_up(inc, off);
    if (lane >= off) inc += y;
  }
  if (lane == 63) s_wave[wave] = inc;
  __syncthreads();
  int wave_off = 0, block_sum = 0;
#pragma unroll
  for (int w = 0; w < kScanThreads / 64; ++w) {
    if (w < wave) wave_off += s_wave[w];
    block_sum += s_wave[w];
  }
  if (wave == 0) {
    // decoupled look-back, 64 predecessors per round: lane i inspects workgroup j - i; the wave adds the aggregates up to
    // the nearest published inclusive prefix (a single thread walking the status words one dependent load at a time cost
    // 30-65 us on the 245-workgroup scans of a training batch)
    long long prefix = 0;
    if (bid > 0) {
      if (lane == 0) st_release(status + bid, kStAgg | (unsigned long long)block_sum);
      int j = bid - 1;
      while (true) {
        const int idx = j - lane;
        const unsigned long long sw = idx >= 0 ? ld_acquire(status + idx) : kStPrefix;      // before workgroup 0: prefix 0
        const unsigned long long state = sw >> 62;
        const unsigned long long empties = __ballot(state < 2), prefixes = __ballot(state == 3);
        const int first_p = prefixes ? (__ffsll((unsigned long long)prefixes) - 1) : 64;
        const unsigned long long window = first_p >= 63 ? ~0ull : ((1ull << (first_p + 1)) - 1);
        if (empties & window) {        // a predecessor inside the window holds its ticket (it is resident) but has not published yet
          __builtin_amdgcn_s_sleep(1);
          continue;
        }
        unsigned long long v = lane <= first_p ? (sw & kStMask) : 0ull;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
          const unsigned lo = __shfl_xor((unsigned)(v & 0xFFFFFFFFull), off), hi = __shfl_xor((unsigned)(v >> 32), off);
          v += ((unsigned long long)hi << 32) | lo;
        }
        prefix += (long long)v;
        if (first_p < 64) break;
        j -= 64;
      }
    }
    if (lane == 0) {
      st_release(status + bid, kStPrefix | (unsigned long long)(prefix + block_sum));
      s_prefix = prefix;
      if (bid == (int)gridDim.x - 1 && total) *total = prefix + block_sum;
    }
  }
  __syncthreads();
  long long ex = s_prefix + wave_off + (inc - tsum);
#pragma unroll
  for (int t = 0; t < kScanItems; ++t) {
    if (base + t < n) out(base + t, ex, v[t]);
    ex += v[t];
  }
}

// The workspaces' 0x7F fill as ONE launch: hipMemsetAsync of a few MB at an arbitrary word count shows up as two fill kernels (aligned body +
// tail), and a training step issues eight of them (three blocks built, two transposed): ~5 us each.  p: 4-byte aligned, nwords 32-bit words.
__global__ __launch_bounds__(256) void fill7f_kernel(uint32_t* __restrict__ p, int64_t nwords) {
  const uint32_t v = 0x7F7F7F7Fu;
  int64_t head = (16 - (int64_t)(reinterpret_cast<uintptr_t>(p) & 15)) & 15;
  head >>= 2;
  if (head > nwords) head = nwords;
  const int64_t body4 = (nwords - head) >> 2, tail0 = head + 4 * body4;
  uint4* __restrict__ q = reinterpret_cast<uint4*>(p + head);
  const int64_t gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = gtid; i < body4; i += stride) q[i] = make_uint4(v, v, v, v);
  if (gtid < head) p[gtid] = v;
  if (gtid < nwords - tail0) p[tail0 + gtid] = v;
}
static int fill7f(void* p, size_t bytes, hipStream_t st, const char* what) {
  if (bytes == 0) return GLNN_OK;
  if ((bytes & 3) || (reinterpret_cast<uintptr_t>(p) & 3)) {                      // (not the layouts of this file; kept correct)
    return hipMemsetAsync(p, 0x7F, bytes, st) == hipSuccess ? GLNN_OK : glnn::fail(GLNN_ERR_HIP, "%s: memset failed", what);
  }
  const int64_t nwords = (int64_t)(bytes >> 2);
  int64_t blocks = (nwords / 4 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(fill7f_kernel, dim3((unsigned)blocks), dim3(256), 0, st, reinterpret_cast<uint32_t*>(p), nwords);
  return glnn::check_launch(what);
}

template <class V, class W>
int launch_scan(V val, int64_t n, unsigned long long* status, int* ticket, W out, int64_t* total, hipStream_t st, const char* what) {
  const int64_t blocks = (n + kScanTile - 1) / kScanTile;
  if (blocks == 0) return GLNN_OK;
  hipLaunchKernelGGL((scan_kernel<V, W>), dim3((unsigned)blocks), dim3(kScanThreads), 0, st, val, n, status, ticket, out, total);
  return glnn::check_launch(what);
}

__device__ __forceinline__ unsigned hash32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ int ht_insert(int* keys, unsigned mask, int key) {
  unsigned slot = hash32((unsigned)key) & mask;
  while (true) {
    const int prev = atomicCAS(&keys[slot], kFill32, key);
    if (prev == kFill32 || prev == key) return (int)slot;
    slot = (slot + 1) & mask;
  }
}

// ------------------------------------------------------------------------------------------------ block builder
struct BlockArgs {
  const int64_t* g_indptr;      // graph CSR (full-neighbour mode), else NULL
  const int32_t* g_indices;
  const int64_t* seeds; int64_t ns;
  const int32_t* smp_src; const int32_t* smp_cnt; int fanout;     // sampled mode, else NULL
  int64_t nnz_cap;              // upper bound of the block's edge count (grid size of the per-edge passes)
  int64_t* indptr;              // out [ns+1]
  int32_t* indices;             // out [nnz_cap] local ids
  int32_t* gindices;            // out [nnz_cap] global ids (optional)
  int64_t* input_nodes;         // out [ns + nnz_cap]
  int64_t* counts;              // out device [2]: nnz, n_src
  int* keys; int* pos; unsigned mask;    // hash table: keys + one position word per slot (mask+1 entries each)
  int* slot_of_edge;            // [nnz_cap]
  int direct;                   // 1: `pos` is indexed by the node id itself (n_nodes entries, no keys array, no probing)
  int n_nodes;                  // direct mode: size of the id universe
  int* bad;                     // direct mode: set to 0 (from the 0x7F fill) by an id outside [0, n_nodes) -> counts[0] = -1
};

// The position word of a node (one per table slot, 0x7F-filled = absent) goes through three stages, all orderable by atomicMin:
//   insert     destination i of the block writes i (< ns); in-edge e writes ns + e  ->  min = the destination index if the node is a
//              destination, else ns + its FIRST edge position (ns + nnz_cap < 2^30: bit 30 is clear in every such value);
//   scan       edge e is a first occurrence of a new source iff pos == ns + e; its dense local id replaces the word with bit 30 set
//              (never equal to any ns + e', so the other edges of that node keep reading "not first");
//   relabel    local id = pos & 0x3FFFFFFF for both kinds.
// One random access per edge and pass (round 4; before: seed_pos / first_pos / local_of_slot, two per edge and pass).
constexpr int kLocalBit = 1 << 30;

// table slot of a node id: the id itself in direct mode (a small id universe: one atomicMin per edge instead of a CAS chain plus
// an atomicMin, and tables of n_nodes instead of 2..4x the frontier entries), else the open-addressing insert
template <bool DIRECT>
__device__ __forceinline__ int slot_of(const BlockArgs& a, int key) {
  if (DIRECT) {
    if ((unsigned)key < (unsigned)a.n_nodes) return key;
    *a.bad = 0;                 // an id outside the universe the caller promised: reported through counts[0], never indexed with
    return 0;
  }
  return ht_insert(a.keys, a.mask, key);
}

struct RowCount {
  BlockArgs a;
  __device__ int operator()(int64_t i) const {
    if (a.smp_cnt) return a.smp_cnt[i];
    const int64_t v = a.seeds[i];
    return (int)(a.g_indptr[v + 1] - a.g_indptr[v]);
  }
};
struct WriteIndptr {
  BlockArgs a;
  __device__ void operator()(int64_t i, long long ex, int v) const {
    a.indptr[i] = ex;
    if (i == a.ns - 1) a.indptr[a.ns] = ex + v;
  }
};

template <bool FULL, bool DIRECT>
__global__ __launch_bounds__(256) void block_insert_kernel(const BlockArgs a) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (FULL) {
    // one wave per row: lanes stride the row's in-edges; wave w < ns also registers seed w
    const int64_t i = t >> 6;
    const int lane = (int)(t & 63);
    if (i >= a.ns) return;
    const int64_t v = a.seeds[i];
    if (lane == 0) {
      const int slot = slot_of<DIRECT>(a, (int)v);
      atomicMin(&a.pos[slot], (int)i);
      a.input_nodes[i] = v;
    }
    const int64_t g0 = a.g_indptr[v], cnt = a.g_indptr[v + 1] - g0, e0 = a.indptr[i];
    for (int64_t k = lane; k < cnt && e0 + k < a.nnz_cap; k += 64) {       // (counts[0] reports the true nnz: the caller checks it against nnz_cap)
      const int key = a.g_indices[g0 + k];
      const int slot = slot_of<DIRECT>(a, key);
      atomicMin(&a.pos[slot], (int)(a.ns + e0 + k));
      a.slot_of_edge[e0 + k] = slot;
      if (a.gindices) a.gindices[e0 + k] = key;
    }
  } else {
    if (t < a.ns) {
      const int64_t v = a.seeds[t];
      const int slot = slot_of<DIRECT>(a, (int)v);
      atomicMin(&a.pos[slot], (int)t);
      a.input_nodes[t] = v;
      return;
    }
    const int64_t s = t - a.ns;
    const int64_t i = s / a.fanout;
    const int k = (int)(s - i * a.fanout);
    if (i >= a.ns || k >= a.smp_cnt[i]) return;
    const int64_t e = a.indptr[i] + k;
    const int key = a.smp_src[i * a.fanout + k];
    const int slot = slot_of<DIRECT>(a, key);
    atomicMin(&a.pos[slot], (int)(a.ns + e));
    a.slot_of_edge[e] = slot;
    if (a.gindices) a.gindices[e] = key;
  }
}

struct FirstSeen {           // 1 iff edge e is the first occurrence of a node that is not a destination of the block
  BlockArgs a;
  __device__ int operator()(int64_t e) const {
    if (e >= a.indptr[a.ns]) return 0;
    return a.pos[a.slot_of_edge[e]] == (int)(a.ns + e) ? 1 : 0;
  }
};
struct AssignLocal {
  BlockArgs a;
  __device__ void operator()(int64_t e, long long ex, int flag) const {
    if (flag) {
      const int slot = a.slot_of_edge[e];
      a.pos[slot] = (int)(a.ns + ex) | kLocalBit;
      a.input_nodes[a.ns + ex] = a.direct ? slot : a.keys[slot];
    }
  }
};

__global__ __launch_bounds__(256) void block_relabel_kernel(const BlockArgs a, const int64_t* n_new) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e == 0) {
    a.counts[0] = (a.direct && *a.bad == 0) ? -1 : a.indptr[a.ns];
    a.counts[1] = a.ns + *n_new;
  }
  if (e >= a.indptr[a.ns]) return;
  a.indices[e] = a.pos[a.slot_of_edge[e]] & (kLocalBit - 1);
}

// GLOBAL-ID block only (indices == NULL && input_nodes == NULL, gindices != NULL): the edges of the block with their global source ids, in the
// order of smp_src / of the graph's rows, behind the row-count scan -- no table, no relabelling (the outermost block of a training batch: its
// consumer gathers from the global feature matrix and never looks at local ids).  counts = {nnz, -1}.
template <bool FULL>
__global__ __launch_bounds__(256) void block_copy_global_kernel(const BlockArgs a) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t == 0) {
    a.counts[0] = a.indptr[a.ns];
    a.counts[1] = -1;
  }
  if (FULL) {
    const int64_t i = t >> 6;
    const int lane = (int)(t & 63);
    if (i >= a.ns) return;
    const int64_t v = a.seeds[i];
    const int64_t g0 = a.g_indptr[v], cnt = a.g_indptr[v + 1] - g0, e0 = a.indptr[i];
    for (int64_t k = lane; k < cnt && e0 + k < a.nnz_cap; k += 64) a.gindices[e0 + k] = a.g_indices[g0 + k];
  } else {
    const int64_t i = t / a.fanout;
    const int k = (int)(t - i * a.fanout);
    if (i >= a.ns || k >= a.smp_cnt[i]) return;
    a.gindices[a.indptr[i] + k] = a.smp_src[i * a.fanout + k];
  }
}

__global__ void block_empty_counts_kernel(int64_t* counts, int64_t* indptr, int64_t ns) {
  counts[0] = 0;
  counts[1] = ns;
  if (indptr) indptr[0] = 0;
}

// ------------------------------------------------------------------------------------------------ CSR transpose
struct TrArgs {
  const int64_t* indptr; const int32_t* indices; int64_t n_dst, n_src, nnz; int add_self;
  int64_t* t_indptr; int32_t* t_indices; int32_t* tmp; int* cursor;
  int cursor_base;      // what the count pass starts from: 0x7F7F7F7F when the cursors were filled by the workspace's ONE memset
};

__global__ __launch_bounds__(256) void tr_count_kernel(const TrArgs a) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < a.nnz; e += stride) atomicAdd(&a.cursor[a.indices[e]], 1);
}
struct TrCount {
  TrArgs a;
  __device__ int operator()(int64_t u) const { return a.cursor[u] - a.cursor_base + ((a.add_self && u < a.n_dst) ? 1 : 0); }
};
struct TrWrite {
  TrArgs a;
  __device__ void operator()(int64_t u, long long ex, int v) const {
    a.t_indptr[u] = ex;
    if (u == a.n_src - 1) a.t_indptr[a.n_src] = ex + v;
    if (a.add_self && u < a.n_dst) {
      a.tmp[ex] = (int)u;          // the self entry u <- u of the SAGE-"gcn" aggregator's backward
      a.cursor[u] = 1;
    } else {
      a.cursor[u] = 0;
    }
  }
};
// one wave per destination row of the ORIGINAL graph: scatter v into the rows of its sources
__global__ __launch_bounds__(256) void tr_fill_kernel(const TrArgs a) {
  const int64_t w = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t v = w; v < a.n_dst; v += n_waves) {
    const int64_t e0 = a.indptr[v], e1 = a.indptr[v + 1];
    for (int64_t e = e0 + lane; e < e1; e += 64) {
      const int u = a.indices[e];
      const int pos = atomicAdd(&a.cursor[u], 1);
      a.tmp[a.t_indptr[u] + pos] = (int)v;
    }
  }
}
// The atomic fill left a row's entries in arbitrary order; order them by value (equal values are interchangeable) so that the fp32 sums
// of the aggregation kernel are reproducible run to run.  A wave takes 64 consecutive rows: rows of <= 4 entries -- nearly all rows of a
// sampled block's transpose (1-2 in-edges per row) -- are ordered by their own lane in registers; longer rows then one at a time by the
// whole wave, ranking by value.  (Until round 5 one wave per row whatever its length: 59 us for the 0.5 M-row transpose of the products
// configuration, most of it the per-row dependent loads.)
__global__ __launch_bounds__(256) void tr_sort_kernel(const TrArgs a) {
  const int64_t w = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t u0 = w * 64; u0 < a.n_src; u0 += n_waves * 64) {
    const int64_t u = u0 + lane;
    const bool valid = u < a.n_src;
    const int64_t my_s0 = valid ? a.t_indptr[u] : 0;
    const int my_len = valid ? (int)(a.t_indptr[u + 1] - my_s0) : 0;
    if (my_len >= 1 && my_len <= 4) {
      int k[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) k[i] = i < my_len ? a.tmp[my_s0 + i] : 0x7FFFFFFF;
#define GLNN_CSWAP(I, J) { const int lo = k[I] < k[J] ? k[I] : k[J], hi = k[I] < k[J] ? k[J] : k[I]; k[I] = lo; k[J] = hi; }
      GLNN_CSWAP(0, 1) GLNN_CSWAP(2, 3) GLNN_CSWAP(0, 2) GLNN_CSWAP(1, 3) GLNN_CSWAP(1, 2)
#undef GLNN_CSWAP
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (i < my_len) a.t_indices[my_s0 + i] = k[i];
    }
    unsigned long long longer = __ballot(my_len > 4);
    while (longer) {
      const int l = __ffsll((unsigned long long)longer) - 1;
      longer &= longer - 1;
      const int64_t s0 = ((int64_t)__shfl((int)(my_s0 >> 32), l) << 32) | (uint32_t)__shfl((int)(my_s0 & 0xFFFFFFFF), l);
      const int len = __shfl(my_len, l);
      const int32_t* src = a.tmp + s0;
      int32_t* dst = a.t_indices + s0;
      if (len <= 64) {
        const int key = lane < len ? src[lane] : 0x7FFFFFFF;
        int rank = 0;
        for (int j = 0; j < len; ++j) {
          const int kj = __shfl(key, j);
          rank += (kj < key || (kj == key && j < lane)) ? 1 : 0;
        }
        if (lane < len) dst[rank] = key;
      } else {
        for (int i = lane; i < len; i += 64) {
          const int key = src[i];
          int rank = 0;
          for (int j = 0; j < len; ++j) {
            const int kj = src[j];
            rank += (kj < key || (kj == key && j < i)) ? 1 : 0;
          }
          dst[rank] = key;
        }
      }
    }
  }
}

}  // namespace

extern "C" int64_t glnn_block_workspace_bytes(int64_t ns, int64_t nnz_cap) {
  if (ns < 0 || nnz_cap < 0) return -1;
  uint64_t cap = 64;
  while (cap < (uint64_t)(2 * (ns + nnz_cap))) cap <<= 1;
  const int64_t scan_words = (ns + kScanTile - 1) / kScanTile + (nnz_cap + kScanTile - 1) / kScanTile + 2;
  return (int64_t)(2 * cap * sizeof(int)) + nnz_cap * (int64_t)sizeof(int) + scan_words * 8 + 64;
}

static int block_build_impl(const int64_t* g_indptr, const int32_t* g_indices, const int64_t* seeds, int64_t ns,
                            const int32_t* smp_src, const int32_t* smp_cnt, int fanout, int64_t nnz_cap,
                            int64_t* indptr, int32_t* indices, int32_t* gindices, int64_t* input_nodes, int64_t* counts,
                            int64_t n_nodes, void* workspace, int64_t workspace_bytes, void* stream) {
  GLNN_REQUIRE(ns >= 0 && nnz_cap >= 0, "glnn_block_build: negative size");
  GLNN_REQUIRE(counts && indptr, "glnn_block_build: null counts/indptr");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (ns == 0 || nnz_cap == 0) {
    if (ns > 0) {
      GLNN_REQUIRE(seeds && (input_nodes || gindices), "glnn_block_build: null pointer");
      if (hipMemsetAsync(indptr, 0, sizeof(int64_t) * (size_t)(ns + 1), st) != hipSuccess ||
          (input_nodes && hipMemcpyAsync(input_nodes, seeds, sizeof(int64_t) * (size_t)ns, hipMemcpyDeviceToDevice, st) != hipSuccess))
        return glnn::fail(GLNN_ERR_HIP, "glnn_block_build: memset/memcpy failed");
    }
    hipLaunchKernelGGL(block_empty_counts_kernel, dim3(1), dim3(1), 0, st, counts, ns == 0 ? indptr : nullptr, ns);
    return glnn::check_launch("glnn_block_build");
  }
  const bool full = smp_src == nullptr;
  const bool global_only = !indices && !input_nodes && gindices;
  GLNN_REQUIRE(seeds && workspace && (global_only || (indices && input_nodes)),
               "glnn_block_build: null pointer (indices and input_nodes may be NULL only together, with gindices: the global-id block)");
  GLNN_REQUIRE(full ? (g_indptr && g_indices && !smp_cnt) : (smp_cnt && fanout >= 1 && nnz_cap >= ns * (int64_t)fanout),
               "glnn_block_build: pass either the graph CSR (full neighbourhood) or smp_src/smp_cnt/fanout with nnz_cap >= ns*fanout");
  GLNN_REQUIRE(ns + nnz_cap < ((int64_t)1 << 30), "glnn_block_build: frontier too large for 32-bit positions");
  GLNN_REQUIRE(workspace_bytes >= glnn_block_workspace_bytes(ns, nnz_cap) && (reinterpret_cast<uintptr_t>(workspace) & 7) == 0,
               "glnn_block_build: workspace needs %lld bytes, 8-byte aligned", (long long)glnn_block_workspace_bytes(ns, nnz_cap));
  uint64_t cap = 64;
  while (cap < (uint64_t)(2 * (ns + nnz_cap))) cap <<= 1;
  // direct mode: every node id is < n_nodes and the id universe is no larger than the hash table's two arrays -- the positions are indexed by
  // the id itself (n_nodes entries: they fit the same workspace), one atomicMin per edge instead of CAS chain + atomicMin, and the
  // 0x7F fill covers n_nodes words instead of 2 cap (the widest block of a products batch: 8 M-slot tables -> a 2.45 M-entry one)
  const bool direct = n_nodes > 0 && (uint64_t)n_nodes <= 2 * cap && n_nodes < kFill32;
  const uint64_t tab = direct ? (uint64_t)((n_nodes + 1) & ~(int64_t)1) : cap;      // entries per table (even: the arrays behind stay 8-byte aligned)
  const int64_t b1 = (ns + kScanTile - 1) / kScanTile, b2 = (nnz_cap + kScanTile - 1) / kScanTile;
  // layout: [status1 b1][status2 b2][ticket1, ticket2 (one 8-byte word)][n_new (8 bytes)][bad (8 bytes)] | (keys,) pos | -- 0x7F-filled up to here
  //         slot_of_edge
  unsigned long long* status1 = reinterpret_cast<unsigned long long*>(workspace);
  unsigned long long* status2 = status1 + b1;
  int* tickets = reinterpret_cast<int*>(status2 + b2);
  int64_t* n_new = reinterpret_cast<int64_t*>(tickets + 2);
  int* bad = reinterpret_cast<int*>(n_new + 1);
  int* tables = reinterpret_cast<int*>(n_new + 2);
  BlockArgs a;
  a.g_indptr = g_indptr; a.g_indices = g_indices; a.seeds = seeds; a.ns = ns; a.smp_src = smp_src; a.smp_cnt = smp_cnt;
  a.fanout = fanout; a.nnz_cap = nnz_cap; a.indptr = indptr; a.indices = indices; a.gindices = gindices; a.input_nodes = input_nodes;
  a.counts = counts; a.direct = direct ? 1 : 0; a.n_nodes = (int)(direct ? n_nodes : 0); a.bad = bad;
  a.keys = direct ? nullptr : tables;
  a.pos = direct ? tables : tables + tab;
  a.mask = (unsigned)(cap - 1); a.slot_of_edge = a.pos + tab;
  const size_t fill_bytes = global_only ? (size_t)((b1 + 1) * 8)       // (the row scan's status words and its ticket)
                                        : (size_t)((b1 + b2 + 3) * 8) + (direct ? 1 : 2) * tab * sizeof(int);
  if (global_only) tickets = reinterpret_cast<int*>(status1 + b1);
  int rc = fill7f(workspace, fill_bytes, st, "glnn_block_build(fill)");
  if (rc != GLNN_OK) return rc;
  rc = launch_scan(RowCount{a}, ns, status1, tickets, WriteIndptr{a}, nullptr, st, "glnn_block_build(scan rows)");
  if (rc != GLNN_OK) return rc;
  if (global_only) {
    const int64_t th = full ? ns * 64 : ns * (int64_t)fanout;
    const dim3 gr((unsigned)((th + 255) / 256));
    if (full) hipLaunchKernelGGL(block_copy_global_kernel<true>, gr, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(block_copy_global_kernel<false>, gr, dim3(256), 0, st, a);
    return glnn::check_launch("glnn_block_build(global ids)");
  }
  const int64_t threads = full ? ns * 64 : ns + ns * (int64_t)fanout;
  const dim3 grid((unsigned)((threads + 255) / 256));
  if (full && direct) hipLaunchKernelGGL((block_insert_kernel<true, true>), grid, dim3(256), 0, st, a);
  else if (full) hipLaunchKernelGGL((block_insert_kernel<true, false>), grid, dim3(256), 0, st, a);
  else if (direct) hipLaunchKernelGGL((block_insert_kernel<false, true>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((block_insert_kernel<false, false>), grid, dim3(256), 0, st, a);
  rc = glnn::check_launch("glnn_block_build(insert)");
  if (rc != GLNN_OK) return rc;
  rc = launch_scan(FirstSeen{a}, nnz_cap, status2, tickets + 1, AssignLocal{a}, n_new, st, "glnn_block_build(scan first occurrences)");
  if (rc != GLNN_OK) return rc;
  hipLaunchKernelGGL(block_relabel_kernel, dim3((unsigned)((nnz_cap + 255) / 256)), dim3(256), 0, st, a, n_new);
  return glnn::check_launch("glnn_block_build(relabel)");
}

extern "C" int glnn_block_build(const int64_t* g_indptr, const int32_t* g_indices, const int64_t* seeds, int64_t ns,
                                const int32_t* smp_src, const int32_t* smp_cnt, int fanout, int64_t nnz_cap,
                                int64_t* indptr, int32_t* indices, int32_t* gindices, int64_t* input_nodes, int64_t* counts,
                                void* workspace, int64_t workspace_bytes, void* stream) {
  return block_build_impl(g_indptr, g_indices, seeds, ns, smp_src, smp_cnt, fanout, nnz_cap, indptr, indices, gindices, input_nodes, counts,
                          0, workspace, workspace_bytes, stream);
}

// ABI 8: the same with the size of the id universe (every seed / neighbour id < n_nodes; 0 = unknown).  When the universe is no larger
// than the hash table would be, the tables are indexed by the id itself (see block_build_impl); results are identical.
extern "C" int glnn_block_build_ids(const int64_t* g_indptr, const int32_t* g_indices, const int64_t* seeds, int64_t ns,
                                    const int32_t* smp_src, const int32_t* smp_cnt, int fanout, int64_t nnz_cap,
                                    int64_t* indptr, int32_t* indices, int32_t* gindices, int64_t* input_nodes, int64_t* counts,
                                    int64_t n_nodes, void* workspace, int64_t workspace_bytes, void* stream) {
  GLNN_REQUIRE(n_nodes >= 0, "glnn_block_build_ids: negative n_nodes");
  return block_build_impl(g_indptr, g_indices, seeds, ns, smp_src, smp_cnt, fanout, nnz_cap, indptr, indices, gindices, input_nodes, counts,
                          n_nodes, workspace, workspace_bytes, stream);
}

extern "C" int64_t glnn_csr_transpose_workspace_bytes(int64_t n_src, int64_t nnz_out) {
  if (n_src < 0 || nnz_out < 0) return -1;
  return ((n_src + kScanTile - 1) / kScanTile + 2) * 8 + (n_src + nnz_out) * (int64_t)sizeof(int) + 64;
}

extern "C" int glnn_csr_transpose(const int64_t* indptr, const int32_t* indices, int64_t n_dst, int64_t n_src, int64_t nnz,
                                  int add_self, int64_t* t_indptr, int32_t* t_indices, void* workspace, int64_t workspace_bytes,
                                  void* stream) {
  GLNN_REQUIRE(n_dst >= 0 && n_src >= 0 && nnz >= 0 && n_src < ((int64_t)1 << 31), "glnn_csr_transpose: bad size");
  GLNN_REQUIRE(t_indptr, "glnn_csr_transpose: null t_indptr");
  GLNN_REQUIRE(!add_self || n_dst <= n_src, "glnn_csr_transpose: add_self needs n_dst <= n_src (a block's destinations are its first sources)");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int64_t nnz_out = nnz + (add_self ? n_dst : 0);
  if (n_src == 0) return hipMemsetAsync(t_indptr, 0, sizeof(int64_t), st) == hipSuccess ? GLNN_OK : glnn::fail(GLNN_ERR_HIP, "glnn_csr_transpose: memset failed");
  GLNN_REQUIRE(indptr && (indices || nnz == 0) && (t_indices || nnz_out == 0) && workspace, "glnn_csr_transpose: null pointer");
  GLNN_REQUIRE(workspace_bytes >= glnn_csr_transpose_workspace_bytes(n_src, nnz_out) && (reinterpret_cast<uintptr_t>(workspace) & 7) == 0,
               "glnn_csr_transpose: workspace needs %lld bytes, 8-byte aligned", (long long)glnn_csr_transpose_workspace_bytes(n_src, nnz_out));
  const int64_t b = (n_src + kScanTile - 1) / kScanTile;
  unsigned long long* status = reinterpret_cast<unsigned long long*>(workspace);
  int* ticket = reinterpret_cast<int*>(status + b);
  int* cursor = ticket + 4;
  TrArgs a;
  a.indptr = indptr; a.indices = indices; a.n_dst = n_dst; a.n_src = n_src; a.nnz = nnz; a.add_self = add_self;
  a.t_indptr = t_indptr; a.t_indices = t_indices; a.cursor = cursor; a.tmp = cursor + n_src;
  // scan status / ticket / per-source counters in ONE fill: the counters start at 0x7F7F7F7F and the scan subtracts it (the write
  // half of the scan re-bases them to 0 / 1 for the fill pass anyway); a second memset is one more ~5 us launch per block
  // per training step.  Degrees that could carry 0x7F7F7F7F + count past INT_MAX keep the separate zero fill.
  const bool one_fill = nnz_out < (1 << 23);
  a.cursor_base = one_fill ? 0x7F7F7F7F : 0;
  {
    const int rcf = fill7f(workspace, (size_t)(b * 8 + 16) + (one_fill ? sizeof(int) * (size_t)n_src : 0), st, "glnn_csr_transpose(fill)");
    if (rcf != GLNN_OK) return rcf;
  }
  if (!one_fill && hipMemsetAsync(cursor, 0, sizeof(int) * (size_t)n_src, st) != hipSuccess)
    return glnn::fail(GLNN_ERR_HIP, "glnn_csr_transpose: memset failed");
  if (nnz > 0) {
    int64_t blocks = (nnz + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(tr_count_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a);
  }
  int rc = launch_scan(TrCount{a}, n_src, status, ticket, TrWrite{a}, nullptr, st, "glnn_csr_transpose(scan)");
  if (rc != GLNN_OK) return rc;
  if (nnz > 0 && n_dst > 0) {
    int64_t blocks = (n_dst * 64 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(tr_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a);
  }
  if (nnz_out > 0) {
    int64_t blocks = (n_src + 255) / 256;                 // 64 rows per wave, 4 waves per workgroup
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(tr_sort_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a);
  }
  return glnn::check_launch("glnn_csr_transpose");
}
