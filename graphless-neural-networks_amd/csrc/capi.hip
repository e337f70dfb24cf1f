// Error state, device query and the K7 row gather / scatter of libglnn_hip.so.
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "glnn_common.h"

namespace glnn {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ---- the library's ONLY read of the environment (see glnn::Options) ----
static long long env_ll(const char* name, long long dflt) {
  const char* e = getenv(name);
  return e ? atoll(e) : dflt;
}
static void load_options(Options& o) {
  o.gemm_pipe = (int)env_ll("GLNN_GEMM_PIPE", 1);
  o.gemm_rowpanel = (int)env_ll("GLNN_GEMM_ROWPANEL", 1);
  o.gemm_lat = (int)env_ll("GLNN_GEMM_LAT", 1);
  o.gemm_tn_lat = (int)env_ll("GLNN_GEMM_TN_LAT", 1);
  o.gemm_tn_lat_splits = (int)env_ll("GLNN_GEMM_TN_LAT_SPLITS", 8);
  o.lat_bn_bwd = (int)env_ll("GLNN_STUDENT_LAT_BN_BWD", 1);
  o.bn_bwd_one_launch = (int)env_ll("GLNN_BN_BWD_ONE_LAUNCH", 1);
  o.narrow_bwd = (int)env_ll("GLNN_STUDENT_NARROW_BWD", 1);
  o.narrow_bwd_min = env_ll("GLNN_STUDENT_NARROW_BWD_MIN", 1ll << 20);
  o.narrow_wgrad = (int)env_ll("GLNN_STUDENT_NARROW_WGRAD", 1);
  o.pad_w0 = (int)env_ll("GLNN_STUDENT_PAD_W0", 1);
  o.slab_consumers = (int)env_ll("GLNN_STUDENT_SLAB_CONSUMERS", 1);
  o.defer_stats = (int)env_ll("GLNN_STUDENT_DEFER_STATS", 1);
  o.batched_wgrad = (int)env_ll("GLNN_STUDENT_BATCHED_WGRAD", 1);
  o.fuse_apply = (int)env_ll("GLNN_STUDENT_FUSE_APPLY", 1);
  o.adam_folds = (int)env_ll("GLNN_STUDENT_ADAM_FOLDS", 1);
  o.gemm_stats = (int)env_ll("GLNN_GEMM_STATS", 1);
  o.sage_fuse_bn_apply = (int)env_ll("GLNN_SAGE_FUSE_BN_APPLY", 1);
  o.spmm_short = (int)env_ll("GLNN_SPMM_SHORT", 1);
  o.sage_fuse_bn_dy = (int)env_ll("GLNN_SAGE_FUSE_BN_DY", 1);
  o.cls_fused = (int)env_ll("GLNN_STUDENT_CLS_FUSED", 1);
  o.signal_fence = env_ll("GLNN_SIGNAL_NO_FENCE", 0) ? 0 : 1;
  o.bn0_in_gemm = (int)env_ll("GLNN_STUDENT_BN0_IN_GEMM", 1);
  o.bn0_consts_in_gemm = (int)env_ll("GLNN_STUDENT_BN0_CONSTS_IN_GEMM", 1);
}
static Options g_opts;
static std::once_flag g_opts_once;
const Options& opts() {
  std::call_once(g_opts_once, [] { load_options(g_opts); });
  return g_opts;
}
}  // namespace glnn

namespace {

// out[i, :] = x[rows[i], :] (gather) or out[rows[i], :] = x[i, :] (scatter); one float4 per thread.
template <bool SCATTER>
__global__ void move_rows_kernel(const float* __restrict__ x, int64_t ldx, const int64_t* __restrict__ rows,
                                 int64_t n_rows, int dv, float* __restrict__ out, int64_t ldo) {
  const int64_t total = n_rows * dv;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / dv;
    const int c4 = (int)(i - r * dv) * 4;
    const int64_t rr = rows[r];
    const int64_t src = SCATTER ? r : rr, dst = SCATTER ? rr : r;
    *reinterpret_cast<float4*>(out + dst * ldo + c4) = *reinterpret_cast<const float4*>(x + src * ldx + c4);
  }
}

template <bool SCATTER>
int move_rows(const float* x, int64_t ldx, const int64_t* rows, int64_t n_rows, int d, float* out, int64_t ldo,
              void* stream, const char* name) {
  GLNN_REQUIRE(d >= 1 && n_rows >= 0, "%s: bad size", name);
  if (n_rows == 0) return GLNN_OK;                      // nothing to move (empty tensors carry null pointers)
  GLNN_REQUIRE(x && rows && out, "%s: null pointer", name);
  const int dpad = (d + 3) & ~3;
  GLNN_REQUIRE(ldx % 4 == 0 && ldo % 4 == 0 && ldx >= dpad && ldo >= dpad, "%s: leading dims must be multiples of 4 and >= %d", name, dpad);
  GLNN_REQUIRE(glnn::aligned16(x) && glnn::aligned16(out), "%s: 16-byte alignment required", name);
  if (n_rows == 0) return GLNN_OK;
  const int dv = dpad / 4;
  int64_t blocks = (n_rows * dv + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL((move_rows_kernel<SCATTER>), dim3((unsigned)blocks), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), x, ldx, rows, n_rows, dv, out, ldo);
  return glnn::check_launch(name);
}

}  // namespace

extern "C" int glnn_abi_version(void) { return 12; }

// test / A-B hook: re-read the GLNN_* switches (glnn::Options).  Not for concurrent use with other calls into the library.
extern "C" void glnn_reload_options(void) {
  (void)glnn::opts();
  glnn::load_options(glnn::g_opts);
}

extern "C" const char* glnn_last_error(void) { return glnn::g_err; }

extern "C" int64_t glnn_struct_bytes(int which) {
  if (which == 0) return (int64_t)sizeof(glnn_mlp_step_desc);
  if (which == 1) return (int64_t)sizeof(glnn_sage_step_desc);
  if (which == 2) return (int64_t)sizeof(glnn_sage_layer);
  if (which == 3) return (int64_t)sizeof(glnn_adam_desc);
  if (which == 4) return (int64_t)sizeof(glnn_hub_plan);
  if (which == 5) return (int64_t)sizeof(glnn_chunk_signals);
  return -1;
}

extern "C" int glnn_device_info(int* cu_count, int* xcd_count, char* arch_buf, int arch_buf_len) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return glnn::fail(GLNN_ERR_NO_DEVICE, "glnn_device_info: no HIP device");
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, dev) != hipSuccess) return glnn::fail(GLNN_ERR_HIP, "glnn_device_info: hipGetDeviceProperties failed");
  if (cu_count) *cu_count = p.multiProcessorCount;
  if (xcd_count) *xcd_count = 8;  // MI355X: 8 XCDs x 32 CUs
  if (arch_buf && arch_buf_len > 0) {
    strncpy(arch_buf, p.gcnArchName, (size_t)arch_buf_len - 1);
    arch_buf[arch_buf_len - 1] = 0;
  }
  return GLNN_OK;
}

extern "C" int glnn_gather_rows_f32(const float* x, int64_t ldx, const int64_t* rows, int64_t n_rows, int d,
                                    float* out, int64_t ldo, void* stream) {
  return move_rows<false>(x, ldx, rows, n_rows, d, out, ldo, stream, "glnn_gather_rows_f32");
}

extern "C" int glnn_scatter_rows_f32(const float* x, int64_t ldx, const int64_t* rows, int64_t n_rows, int d,
                                     float* out, int64_t ldo, void* stream) {
  return move_rows<true>(x, ldx, rows, n_rows, d, out, ldo, stream, "glnn_scatter_rows_f32");
}
