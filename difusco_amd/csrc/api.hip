// C ABI of libdifusco_hip.so (declared in include/difusco_hip.h) and the step driver that strings the
// gfx950 kernels into one reverse-diffusion step.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <vector>

#include "../../include/difusco_hip.h"
#include "kernels.h"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

#define HIP_TRY(expr)                                                                         \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess) return fail(DIFUSCO_EHIP, "%s: %s", #expr, hipGetErrorString(e_));  \
  } while (0)

}  // namespace

namespace difusco {
// error reporting for the other translation units of the library (same thread-local message buffer)
int set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
}  // namespace difusco

namespace {

constexpr int64_t kAlignFloats = 64;  // every packed tensor starts on a 256-byte boundary

int64_t align_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }

struct Layout {
  std::vector<int64_t> off;
  int64_t total = 0;
  int64_t layer_stride = 0;  // floats between the same tensor of consecutive layers
};

// hipMemsetAsync of zero bytes is an error while a stream is being captured into a hipGraph
hipError_t zero_async(void* p, size_t bytes, hipStream_t st) { return bytes ? hipMemsetAsync(p, 0, bytes, st) : hipSuccess; }

bool hidden_ok(int h) { return h == 64 || h == 128 || h == 256; }

Layout make_layout(int H, int L, int C) {
  Layout lo;
  const int64_t T2 = H / 2;
  int64_t cur = 0;
  auto push = [&](int64_t n) {
    lo.off.push_back(cur);
    cur = align_up(cur + n, kAlignFloats);
  };
  push((int64_t)H * H); push(H);          // node_embed
  push((int64_t)H * H); push(H);          // edge_embed
  push(T2 * H); push(T2);                 // time_embed.0
  push(T2 * T2); push(T2);                // time_embed.2
  push(H); push(H);                       // out.0 (GroupNorm affine)
  push((int64_t)C * H); push(C);          // out.2 (1x1 conv)
  push(T2); push(T2); push(H);            // freqs, dimt_pos, dimt_scalar
  push((int64_t)5 * H * H / 2 + H);       // edge_embed split planes (3 bf16 + 2 fp16 planes of H*H) + fp16 row inverse scales
  const int64_t layer0 = cur;
  for (int l = 0; l < L; ++l) {
    push((int64_t)4 * H * H); push((int64_t)4 * H);  // node4 = U|V|A|B
    push((int64_t)H * H); push(H);                    // C
    push(H); push(H); push(H); push(H);               // norm_h, norm_e
    push((int64_t)H * T2); push(H);                   // time layer
    push(H); push(H);                                 // per_layer_out LN
    push((int64_t)H * H); push(H);                    // per_layer_out linear
    push((int64_t)5 * H * H / 2 + H); push((int64_t)5 * H * H / 2 + H);  // split planes of C and per_layer_out (+ inverse scales)
    push((int64_t)5 * 4 * H * H / 2 + 4 * H);                             // split planes of the node linear (+ inverse scales)
    push(8);                                                               // operand scales of the fused edge kernel
    push((int64_t)4 * H); push((int64_t)8 * H);                            // node linear, fused path: bias, column scales (fp16 | bf16 planes)
    if (l == 0) lo.layer_stride = cur - layer0;
  }
  lo.total = cur;
  return lo;
}

struct Workspace {
  float *h, *node4, *e, *tmp, *tbias, *table_in, *table, *stats, *part, *direct, *gn_tile, *etmax, *hscale, *escale;
  double* partial;
  size_t bytes;
};

// pieces of the neighbour sum written by the fused edge kernel: 2 per 32-edge tile (8 tiles per workgroup)
size_t fused_part_floats(int64_t E) { return (size_t)((E + 255) / 256) * 8 * 2 * 256; }

Workspace carve(void* base, int H, int L, int64_t N, int64_t E, int S, int nblk) {
  Workspace w;
  size_t cur = 0;
  auto take = [&](size_t bytes) {
    size_t at = cur;
    cur = (cur + bytes + 255) / 256 * 256;
    return base ? (void*)((char*)base + at) : (void*)nullptr;
  };
  w.h = (float*)take(sizeof(float) * N * H);
  w.node4 = (float*)take(sizeof(float) * N * 4 * H);
  w.e = (float*)take(sizeof(float) * (H == 256 ? (E + 255) / 256 * 256 : E) * H);   // fused path: tiles of 256 edges
  w.tmp = (float*)take(sizeof(float) * (E > 2 ? E : 2) * H);
  w.tbias = (float*)take(sizeof(float) * L * H);
  w.table_in = (float*)take(sizeof(float) * 2 * H);
  w.table = (float*)take(sizeof(float) * 4 * H);      // rows 0,1: edge-input table; rows 2,3: C of layer 0 applied to them
  w.stats = (float*)take(sizeof(float) * S * 64);
  w.partial = (double*)take(sizeof(double) * (size_t)S * (nblk < 256 ? 256 : nblk) * 64);
  w.part = (float*)take(H == 256 ? sizeof(float) * fused_part_floats(E) : 0);
  w.direct = (float*)take(H == 256 ? sizeof(float) * N * H : 0);
  w.gn_tile = (float*)take(H == 256 ? sizeof(float) * ((E + 255) / 256 * 8) * 64 : 0);   // per 32-edge tile: 32 x (sum, sumsq)
  // operand scales of the fp16 split path: max |e| per 32-edge tile (fused kernel), one power of two per node row / edge row
  w.etmax = (float*)take(H == 256 ? sizeof(float) * ((E + 255) / 256 * 8) : 0);
  w.hscale = (float*)take(sizeof(float) * N);
  w.escale = (float*)take(sizeof(float) * E);
  w.bytes = cur;
  return w;
}


// Prepared state of a TSP call shape (difusco_step_args.prepared): what a step computes from (weights, points) alone.
struct Prepared {
  float *h0, *node4_0, *table;      // [N,H], [N,4H], [4,H]
  size_t bytes;
};
Prepared carve_prepared(void* base, int H, int64_t N) {
  Prepared p;
  size_t cur = 0;
  auto take = [&](size_t bytes) {
    size_t at = cur;
    cur = (cur + bytes + 255) / 256 * 256;
    return base ? (float*)((char*)base + at) : (float*)nullptr;
  };
  p.h0 = take(sizeof(float) * N * H);
  p.node4_0 = take(sizeof(float) * N * 4 * H);
  p.table = take(sizeof(float) * 4 * H);
  p.bytes = cur;
  return p;
}

// ---- optional in-library profiler: HIP events around every kernel launch, per category ----------
// (bench.py needs the average duration of the dominant kernel measured on the launch stream inside
// the timed region; the launches happen inside difusco_denoise_step, so the brackets live here.)
enum { PROF_LINEAR_EDGE = 0, PROF_LINEAR_NODE, PROF_GATE, PROF_HEAD, PROF_EMBED, PROF_NCAT };
// (process-wide, like the HIP events it owns; every access goes through g_prof_mu, so that two engines driven from two
// host threads cannot corrupt it - their brackets simply share the category totals)
std::mutex g_prof_mu;
struct Profiler {
  bool on = false;
  bool dominant_only = false;   // bracket category 0 (the E-row / fused edge-layer launches) only
  std::vector<hipEvent_t> ev;   // pairs
  std::vector<int> cat;
  size_t used = 0;              // pairs in use
} g_prof;

struct ProfScope {
  hipStream_t st;
  bool active;
  hipEvent_t end;      // handle copied while the lock is held: the destructor never touches the shared vectors
  ProfScope(int category, hipStream_t s) : st(s), active(false), end(nullptr) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (!g_prof.on || (g_prof.dominant_only && category != PROF_LINEAR_EDGE) || g_prof.used * 2 + 2 > g_prof.ev.size()) return;
    const size_t slot = g_prof.used++;
    g_prof.cat[slot] = category;
    end = g_prof.ev[2 * slot + 1];
    active = hipEventRecord(g_prof.ev[2 * slot], st) == hipSuccess;
  }
  ~ProfScope() {
    if (active) (void)hipEventRecord(end, st);
  }
};

}  // namespace

#ifdef DIFUSCO_PROFILING
static int g_step_skip = 0;      // difusco_debug_set key 11 (profiling library): launches of the step that are skipped for an upper-bound timing
#endif

extern "C" {

int difusco_abi_version(void) { return DIFUSCO_ABI_VERSION; }

const char* difusco_last_error(void) { return g_err; }

int difusco_weights_layout(int hidden, int n_layers, int out_channels, int64_t* offsets, int max_entries,
                           int64_t* total_floats) {
  if (!hidden_ok(hidden)) return fail(DIFUSCO_EINVAL, "hidden must be 64, 128 or 256 (got %d)", hidden);
  if (n_layers < 1 || (out_channels != 1 && out_channels != 2))
    return fail(DIFUSCO_EINVAL, "n_layers >= 1 and out_channels in {1,2} required");
  Layout lo = make_layout(hidden, n_layers, out_channels);
  const int n = (int)lo.off.size();
  if (offsets) {
    if (max_entries < n) return fail(DIFUSCO_EINVAL, "offset array too small: need %d", n);
    std::memcpy(offsets, lo.off.data(), sizeof(int64_t) * n);
  }
  if (total_floats) *total_floats = lo.total;
  return n;
}

int difusco_csr_from_coo_host(const int64_t* edge_index, int64_t n_edges, int64_t n_nodes, int32_t* rowptr,
                              int32_t* col, int32_t* row, int32_t* perm, int* identity) {
  if (!edge_index || !rowptr || !col || !row || !perm) return fail(DIFUSCO_EINVAL, "null pointer");
  if (n_edges < 0 || n_nodes < 0 || n_edges > INT32_MAX || n_nodes >= INT32_MAX)
    return fail(DIFUSCO_EINVAL, "sizes out of int32 range");
  const int64_t* r = edge_index;
  const int64_t* c = edge_index + n_edges;
  for (int64_t i = 0; i <= n_nodes; ++i) rowptr[i] = 0;
  for (int64_t k = 0; k < n_edges; ++k) {
    if (r[k] < 0 || r[k] >= n_nodes || c[k] < 0 || c[k] >= n_nodes)
      return fail(DIFUSCO_EINVAL, "edge %lld = (%lld,%lld) out of range [0,%lld)", (long long)k, (long long)r[k],
                  (long long)c[k], (long long)n_nodes);
    rowptr[r[k] + 1]++;
  }
  for (int64_t i = 0; i < n_nodes; ++i) rowptr[i + 1] += rowptr[i];
  std::vector<int32_t> fill(rowptr, rowptr + n_nodes);
  int ident = 1;
  for (int64_t k = 0; k < n_edges; ++k) {  // stable counting sort by centre node
    const int32_t s = fill[r[k]]++;
    col[s] = (int32_t)c[k];
    row[s] = (int32_t)r[k];
    perm[s] = (int32_t)k;
    if (s != k) ident = 0;
  }
  if (identity) *identity = ident;
  return DIFUSCO_OK;
}

size_t difusco_workspace_bytes(int hidden, int n_layers, int n_nodes, int n_edges, int n_segments) {
  if (!hidden_ok(hidden) || n_layers < 1 || n_nodes < 0 || n_edges < 0 || n_segments < 1) return 0;
  const long long rows = n_edges > n_nodes ? n_edges : n_nodes;
  return carve(nullptr, hidden, n_layers, n_nodes, n_edges, n_segments, difusco::gn_blocks_for(rows)).bytes;
}

int difusco_denoise_step(const difusco_step_args* a) {
  using namespace difusco;
  if (!a) return fail(DIFUSCO_EINVAL, "null args");
  if (a->struct_size != sizeof(difusco_step_args) || a->abi_version != DIFUSCO_ABI_VERSION)
    return fail(DIFUSCO_EINVAL, "ABI mismatch: struct_size %u (want %zu), abi %u (want %d)", a->struct_size,
                sizeof(difusco_step_args), a->abi_version, DIFUSCO_ABI_VERSION);
  const int H = a->hidden, L = a->n_layers, C = a->out_channels;
  if (!hidden_ok(H)) return fail(DIFUSCO_EINVAL, "hidden must be 64, 128 or 256 (got %d)", H);
  if (L < 1) return fail(DIFUSCO_EINVAL, "n_layers must be >= 1");
  if (a->task != DIFUSCO_TASK_TSP && a->task != DIFUSCO_TASK_MIS) return fail(DIFUSCO_EINVAL, "unknown task %d", a->task);
  if (a->diffusion != DIFUSCO_CATEGORICAL && a->diffusion != DIFUSCO_GAUSSIAN)
    return fail(DIFUSCO_EINVAL, "unknown diffusion %d", a->diffusion);
  if (C != (a->diffusion == DIFUSCO_CATEGORICAL ? 2 : 1))
    return fail(DIFUSCO_EINVAL, "out_channels %d does not match diffusion type", C);
  const int64_t N = a->n_nodes, E = a->n_edges;
  if (N <= 0 || E < 0) return fail(DIFUSCO_EINVAL, "n_nodes must be > 0, n_edges >= 0");
  if (!a->weights || !a->rowptr || (E > 0 && !a->col) || !a->xt || !a->xt_out || !a->workspace)
    return fail(DIFUSCO_EINVAL, "null device pointer (weights/rowptr/col/xt/xt_out/workspace)");
  if (a->task == DIFUSCO_TASK_TSP && !a->points && !a->prepared) return fail(DIFUSCO_EINVAL, "TSP needs points");
  if (a->task == DIFUSCO_TASK_TSP && E == 0) return fail(DIFUSCO_EINVAL, "TSP needs edges");
  if (a->n_segments < 1 || (a->n_segments > 1 && !a->seg_ptr))
    return fail(DIFUSCO_EINVAL, "n_segments >= 1, seg_ptr required when > 1");
  if (a->rand_mode == DIFUSCO_RAND_INJECTED && !a->rand) return fail(DIFUSCO_EINVAL, "injected randomness needs rand");
  if (a->rand_mode < 0 || a->rand_mode > 2) return fail(DIFUSCO_EINVAL, "unknown rand_mode %d", a->rand_mode);
  if (a->gn_phase < 0 || a->gn_phase > 2) return fail(DIFUSCO_EINVAL, "gn_phase must be 0, 1 or 2");
  if (a->aggregation < DIFUSCO_AGG_SUM || a->aggregation > DIFUSCO_AGG_MAX)
    return fail(DIFUSCO_EINVAL, "unknown aggregation %d (DIFUSCO_AGG_SUM / MEAN / MAX)", a->aggregation);
  if (a->gn_phase != 0 && (a->n_segments != 1 || !a->gn_sums))
    return fail(DIFUSCO_EINVAL, "gn_phase %d needs n_segments == 1 and a gn_sums buffer of 65 doubles", a->gn_phase);
  const bool head_only = a->gn_phase == 2;   // everything up to the statistics was done by the phase-1 call
  const bool needs_draw = a->post[4] != 0.0f;
  if (needs_draw && a->rand_mode == DIFUSCO_RAND_NONE)
    return fail(DIFUSCO_EINVAL, "this step draws random numbers: rand_mode must not be NONE");

  const bool tsp = a->task == DIFUSCO_TASK_TSP;
  const int64_t out_rows = tsp ? E : N;
  const int nblk = gn_blocks_for(E > N ? E : N);
  Workspace ws = carve(a->workspace, H, L, N, E, a->n_segments, nblk);
  if (ws.bytes > a->workspace_bytes)
    return fail(DIFUSCO_EWORKSPACE, "workspace too small: %zu < %zu", a->workspace_bytes, ws.bytes);

  const Layout lo = make_layout(H, L, C);
  const float* W = a->weights;
  auto G = [&](int id) { return W + lo.off[id]; };
  auto LW = [&](int l, int id) { return W + lo.off[DIFUSCO_W_GLOBAL_COUNT + l * DIFUSCO_WL_COUNT + id]; };
  hipStream_t st = (hipStream_t)a->stream;
  if (a->precision < DIFUSCO_PREC_FP32 || a->precision > DIFUSCO_PREC_FP16X3)
    return fail(DIFUSCO_EINVAL, "unknown precision %d", a->precision);
  // E-row linear: exact fp32 MFMA or bf16 split planes, same contract
  auto edge_linear = [&](const float* x, const float* w, const float* planes, const float* b, const float* res,
                         float* y) -> hipError_t {
    if (a->precision == DIFUSCO_PREC_FP32) return linear_rows(x, w, b, res, y, E, H, H, H, st);
    const unsigned short* pl = reinterpret_cast<const unsigned short*>(planes);
    SplitScale sc;
    if (a->precision == DIFUSCO_PREC_FP16X3) {
      pl += (long long)3 * H * H;   // fp16 planes follow the 3 bf16 planes
      // fp16 planes: per-row power-of-two scale of x (one extra pass over x on this unfused path), weight inverse scales
      hipError_t er = launch_row_pow2_scale(x, E, H, ws.escale, st);
      if (er != hipSuccess) return er;
      sc.row_scale = ws.escale;
      sc.w_inv = planes + (long long)5 * H * H / 2;
    }
    return linear_rows_split(x, pl, (long long)H * H, a->precision, b, res, y, E, H, H, H, st, 0, sc);
  };

  // PROF(category, call): HIP_TRY(call), bracketed by a pair of HIP events on `st` when profiling is on
#define PROF(cat, call)      \
  {                          \
    ProfScope ps_(cat, st);  \
    HIP_TRY(call);           \
  }

  // fused edge-layer path: H = 256, 16-bit split planes, one GroupNorm statistic segment; e is then kept in the
  // tiled layout (kernels.h: edge_tiled_offset) from the embedding to the head
  // (per-sample statistic segments - the dense mode of TSP-50 / 100 with parallel_sampling > 1 - run the same fused layers;
  // only the head differs: masked per-segment statistics and a per-row segment look-up, launch_head_tiled)
  // the full-line neighbour-table gathers address node rows by 32-bit byte offsets (4 KB per node): calls with 2^20 nodes or
  // more take the register-gather instantiation of the same kernel (64-bit addresses, bit-identical results, ~3 % slower)
  const int reg_gather = N >= (1 << 20) ? 1 : 0;
  // aggregation = max has fused instantiations of its own (the per-tile pieces are maxima) for the full-line gather form only:
  // with 2^20 nodes or more it takes the unfused sequence
  const bool agg_max = a->aggregation == DIFUSCO_AGG_MAX;
  const bool fused = H == 256 && !a->no_fusion && E > 0 && !(agg_max && reg_gather) &&
                     (a->precision == DIFUSCO_PREC_BF16X3 || a->precision == DIFUSCO_PREC_FP16X3);
  if (fused && !a->row) return fail(DIFUSCO_EINVAL, "the fused edge-layer kernel needs args->row");
  const int fused_variant = reg_gather | (agg_max ? 2 : 0);      // (launch_by_mode, edge_layer.hip)
  const int64_t E_pad = (E + 255) / 256 * 256;
  // first layer: when the edge input is a table lookup (categorical TSP: embedding of the bit; MIS: zeros) the fused
  // kernel takes it from the table and the pass that would write e0 to HBM is skipped
  // last layer (at least two layers).  TSP: the fused kernel also emits the head's GroupNorm partial sums per tile and
  // skips the node update nobody reads; MIS: it skips the edge output nobody reads.
#ifdef DIFUSCO_PROFILING
  const bool ablating = difusco::g_fused_ablate != 0;
#else
  const bool ablating = false;
#endif
  const bool tail_fold = fused && L >= 2 && (a->flags & DIFUSCO_FLAG_NO_TAIL_FOLD) == 0 && !ablating;
  const bool gn_fold = tail_fold && tsp;
  const bool l0_fold = fused && (a->flags & DIFUSCO_FLAG_NO_L0_FOLD) == 0 && !ablating &&
                       (tsp ? a->xt_is_binary != 0 : true);
  const bool f16 = a->precision == DIFUSCO_PREC_FP16X3;
  const int64_t n_tiles_pad = E_pad / 32;
  // prepared state (TSP, fused path only; anything else recomputes - the stateless step is always correct)
  const bool use_prep = a->prepared != nullptr && fused && tsp && !ablating;
  const Prepared prep = carve_prepared(const_cast<void*>(a->prepared), H, N);
  const float* table = use_prep ? prep.table : ws.table;      // two-row edge-input table (+ C of layer 0 applied to it)
  if (tsp && !a->points && !use_prep && !head_only)
    return fail(DIFUSCO_EINVAL, "TSP needs points (only a fused-path step with prepared state runs without them)");

  // per-layer time bias rows: time_layer_l(time_embed(timestep_embedding(t)))   [L,H]
  const float* tbias = a->tbias ? a->tbias : ws.tbias;      // (caller-prepared rows of this t, or computed here)
  if (!head_only && !a->tbias)
  PROF(PROF_EMBED, launch_time_bias(&a->t, 1, H, L, G(DIFUSCO_W_TIME_FREQS), G(DIFUSCO_W_TIME0_W), G(DIFUSCO_W_TIME0_B),
                                    G(DIFUSCO_W_TIME2_W), G(DIFUSCO_W_TIME2_B), LW(0, 0), lo.layer_stride,
                                    lo.off[DIFUSCO_W_GLOBAL_COUNT + DIFUSCO_WL_TIME_W] - lo.off[DIFUSCO_W_GLOBAL_COUNT],
                                    lo.off[DIFUSCO_W_GLOBAL_COUNT + DIFUSCO_WL_TIME_B] - lo.off[DIFUSCO_W_GLOBAL_COUNT],
                                    ws.tbias, st))

  // input embeddings (gnn_encoder.py:394-395 TSP, :405-407 MIS).  node4 doubles as scratch for the
  // sinusoidal node features before the first layer overwrites it.
  if (head_only) {
    // (phase 2: e / h of the phase-1 call are still in the workspace)
  } else if (tsp) {
    if (!use_prep) {
      PROF(PROF_EMBED, launch_pos_embed(a->points, G(DIFUSCO_W_DIMT_POS), (int)N, H, ws.node4, st))
      PROF(PROF_LINEAR_NODE, linear_rows(ws.node4, G(DIFUSCO_W_NODE_EMBED_W), G(DIFUSCO_W_NODE_EMBED_B), nullptr, ws.h, N,
                                         H, H, H, st))
    }
    if (fused)   // pad lanes of the last tiles must read as zero in every later kernel
      PROF(PROF_EMBED, zero_async(ws.e + (E / 32) * 32 * H, sizeof(float) * (E_pad - (E / 32) * 32) * H, st))
    if (a->xt_is_binary) {
      if (!use_prep) {
        PROF(PROF_EMBED, launch_scalar_embed(nullptr, nullptr, G(DIFUSCO_W_DIMT_SCALAR), 2, H, ws.table_in, st))
        PROF(PROF_EMBED, launch_two_rows_linear(H, ws.table_in, G(DIFUSCO_W_EDGE_EMBED_W), G(DIFUSCO_W_EDGE_EMBED_B), ws.table,
                                                st))
      }
      if (l0_fold) {   // e0 and C e0 are read from the table by the first fused layer: C on the two rows, exact fp32
        if (!use_prep)
          PROF(PROF_EMBED, launch_two_rows_linear(H, ws.table, LW(0, DIFUSCO_WL_C_W), nullptr, ws.table + 2 * H, st))
      }
      else if (fused) {
        PROF(PROF_EMBED, launch_table_rows_tiled(a->xt, a->perm, table, E, ws.e, st))
        if (f16) PROF(PROF_EMBED, launch_tile_absmax_tiled(ws.e, n_tiles_pad, ws.etmax, st))
      }
      else PROF(PROF_EMBED, launch_table_rows(a->xt, a->perm, table, E, H, ws.e, st))
    } else {
      if (fused) {   // sinusoidal features generated inside the linear: the E x H embedding never exists in memory
        const unsigned short* pl = reinterpret_cast<const unsigned short*>(G(DIFUSCO_W_EDGE_EMBED_PLANES)) +
                                   (a->precision == DIFUSCO_PREC_FP16X3 ? (long long)3 * H * H : 0);
        PROF(PROF_EMBED, launch_edge_embed_tiled(a->xt, a->perm, G(DIFUSCO_W_DIMT_SCALAR), pl, (long long)H * H, a->precision,
                                                 f16 ? G(DIFUSCO_W_EDGE_EMBED_PLANES) + (long long)5 * H * H / 2 : nullptr,
                                                 G(DIFUSCO_W_EDGE_EMBED_B), ws.e, E, f16 ? ws.etmax : nullptr, st, a->gen_table,
                                                 a->gen_table ? reinterpret_cast<int*>(ws.escale) : nullptr))      // (escale: unfused path only - free here)
      } else {
        PROF(PROF_EMBED, launch_scalar_embed(a->xt, a->perm, G(DIFUSCO_W_DIMT_SCALAR), E, H, ws.tmp, st))
        PROF(PROF_LINEAR_EDGE, edge_linear(ws.tmp, G(DIFUSCO_W_EDGE_EMBED_W), G(DIFUSCO_W_EDGE_EMBED_PLANES),
                                           G(DIFUSCO_W_EDGE_EMBED_B), nullptr, ws.e))
      }
    }
  } else {
    PROF(PROF_EMBED, launch_scalar_embed(a->xt, nullptr, G(DIFUSCO_W_DIMT_SCALAR), N, H, ws.node4, st))
    PROF(PROF_LINEAR_NODE, linear_rows(ws.node4, G(DIFUSCO_W_NODE_EMBED_W), G(DIFUSCO_W_NODE_EMBED_B), nullptr, ws.h, N,
                                       H, H, H, st))
    if (l0_fold) {   // e = zeros (gnn_encoder.py:407) comes from an all-zero table; only the pad tail must read as zero
      PROF(PROF_EMBED, hipMemsetAsync(ws.table, 0, sizeof(float) * 4 * H, st))
      PROF(PROF_EMBED, zero_async(ws.e + (E / 32) * 32 * H, sizeof(float) * (E_pad - (E / 32) * 32) * H, st))
    } else if (E > 0) {
      PROF(PROF_EMBED, hipMemsetAsync(ws.e, 0, sizeof(float) * (fused ? E_pad : E) * H, st))
      if (fused && f16) PROF(PROF_EMBED, hipMemsetAsync(ws.etmax, 0, sizeof(float) * n_tiles_pad, st))
    }
  }

  // the GNN layers (gnn_encoder.py:425-449)
  const long long split_off = a->precision == DIFUSCO_PREC_FP16X3 ? (long long)3 * H * H : 0;  // fp16 planes follow bf16
  for (int l = 0; l < (head_only ? 0 : L); ++l) {
    // layer 0 of a step with prepared state: U|V|A|B of h0 and h0 itself come from the prepared buffer
    const bool prep_l = use_prep && l == 0;
    const float* node4 = prep_l ? prep.node4_0 : ws.node4;
    if (prep_l) {
    } else
#ifdef DIFUSCO_PROFILING
    if ((g_step_skip & 2) && l >= 2) {      // (timing only, WRONG results: the node linear's launches from layer 2 on are skipped)
    } else
#endif
    if (a->precision != DIFUSCO_PREC_FP32 && H == 256) {   // node rows on the same split-precision matrix-core path
      const unsigned short* npl = reinterpret_cast<const unsigned short*>(LW(l, DIFUSCO_WL_NODE4_PLANES)) +
                                  (a->precision == DIFUSCO_PREC_FP16X3 ? (long long)3 * 4 * H * H : 0);
      SplitScale sc;
      if (f16) {   // row scales of h: left by node_finalize of the previous layer on the fused path, computed here otherwise
        if (!fused || l == 0) PROF(PROF_LINEAR_NODE, launch_row_pow2_scale(ws.h, N, H, ws.hscale, st))
        sc.row_scale = ws.hscale;
        sc.w_inv = LW(l, DIFUSCO_WL_NODE4_PLANES) + (long long)5 * 4 * H * H / 2;
      }
      // fused edge kernel: A | B rows in its log2(e) domain, b_C folded into the A rows (difusco_hip.h, ABI 11 note)
      if (fused) sc.w_inv = LW(l, DIFUSCO_WL_NODE4_FUSED_S) + (f16 ? 0 : 4 * H);
      PROF(PROF_LINEAR_NODE, linear_rows_split(ws.h, npl, (long long)4 * H * H, a->precision,
                                               LW(l, fused ? DIFUSCO_WL_NODE4_FUSED_B : DIFUSCO_WL_NODE4_B),
                                               nullptr, ws.node4, N, H, 4 * H, 4 * H, st, 0, sc))
    } else {
      PROF(PROF_LINEAR_NODE, linear_rows(ws.h, LW(l, DIFUSCO_WL_NODE4_W), LW(l, DIFUSCO_WL_NODE4_B), nullptr, ws.node4,
                                         N, H, 4 * H, 4 * H, st))
    }
    if (fused && l == 0 && l0_fold) {
      PROF(PROF_LINEAR_EDGE,
           launch_edge_layer_fused_l0(a->precision, ws.e, node4, a->row, a->col, (int)E,
                                      reinterpret_cast<const unsigned short*>(LW(l, DIFUSCO_WL_C_PLANES)) + split_off,
                                      reinterpret_cast<const unsigned short*>(LW(l, DIFUSCO_WL_OUT_PLANES)) + split_off,
                                      (long long)H * H, LW(l, DIFUSCO_WL_C_B), LW(l, DIFUSCO_WL_NORM_E_W),
                                      LW(l, DIFUSCO_WL_NORM_E_B), tbias + (size_t)l * H, LW(l, DIFUSCO_WL_OUT_LN_W),
                                      LW(l, DIFUSCO_WL_OUT_LN_B), LW(l, DIFUSCO_WL_OUT_B), tsp ? 1 : 0, ws.part, ws.direct,
                                      table, tsp ? a->xt : nullptr, tsp ? a->perm : nullptr,
                                      LW(l, DIFUSCO_WL_FUSED_SCALES), ws.etmax, st, fused_variant))
    } else if (fused && tail_fold && l == L - 1) {
      PROF(PROF_LINEAR_EDGE,
           launch_edge_layer_fused_tail(a->precision, tsp ? 1 : 2, ws.e, node4, a->row, a->col, (int)E,
                                      reinterpret_cast<const unsigned short*>(LW(l, DIFUSCO_WL_C_PLANES)) + split_off,
                                      reinterpret_cast<const unsigned short*>(LW(l, DIFUSCO_WL_OUT_PLANES)) + split_off,
                                      (long long)H * H, LW(l, DIFUSCO_WL_C_B), LW(l, DIFUSCO_WL_NORM_E_W),
                                      LW(l, DIFUSCO_WL_NORM_E_B), tbias + (size_t)l * H, LW(l, DIFUSCO_WL_OUT_LN_W),
                                      LW(l, DIFUSCO_WL_OUT_LN_B), LW(l, DIFUSCO_WL_OUT_B), tsp ? 1 : 0, ws.part, ws.direct,
                                      ws.gn_tile, LW(l, DIFUSCO_WL_FUSED_SCALES), ws.etmax, st, fused_variant))
    } else if (fused) {
      PROF(PROF_LINEAR_EDGE,
           launch_edge_layer_fused(a->precision, ws.e, node4, a->row, a->col, (int)E,
                                   reinterpret_cast<const unsigned short*>(LW(l, DIFUSCO_WL_C_PLANES)) + split_off,
                                   reinterpret_cast<const unsigned short*>(LW(l, DIFUSCO_WL_OUT_PLANES)) + split_off,
                                   (long long)H * H, LW(l, DIFUSCO_WL_C_B), LW(l, DIFUSCO_WL_NORM_E_W),
                                   LW(l, DIFUSCO_WL_NORM_E_B), tbias + (size_t)l * H, LW(l, DIFUSCO_WL_OUT_LN_W),
                                   LW(l, DIFUSCO_WL_OUT_LN_B), LW(l, DIFUSCO_WL_OUT_B), tsp ? 1 : 0, ws.part, ws.direct,
                                   LW(l, DIFUSCO_WL_FUSED_SCALES), ws.etmax, ws.etmax, st, fused_variant))
    }
    if (fused && gn_fold && l == L - 1) continue;     // TSP: h is not read after the last layer
    if (fused) {
#ifdef DIFUSCO_PROFILING
      if (g_step_skip & 1) continue;      // (timing only, WRONG results: what any fold of node_finalize could save at most)
#endif
      PROF(PROF_GATE, launch_node_finalize((int)N, (int)E, a->rowptr, node4, ws.part, ws.direct, ws.h,
                                           LW(l, DIFUSCO_WL_NORM_H_W), LW(l, DIFUSCO_WL_NORM_H_B),
                                           tbias + (size_t)l * H, tsp ? 1 : 0, f16 ? ws.hscale : nullptr, st,
                                           prep_l ? prep.h0 : nullptr, a->aggregation))
      continue;
    }
    PROF(PROF_LINEAR_EDGE, edge_linear(ws.e, LW(l, DIFUSCO_WL_C_W), LW(l, DIFUSCO_WL_C_PLANES), LW(l, DIFUSCO_WL_C_B),
                                       nullptr, ws.tmp))
    PROF(PROF_GATE, launch_edge_gate_aggregate(H, (int)N, a->rowptr, a->col, ws.node4, ws.tmp, ws.h,
                                               LW(l, DIFUSCO_WL_NORM_H_W), LW(l, DIFUSCO_WL_NORM_H_B),
                                               LW(l, DIFUSCO_WL_NORM_E_W), LW(l, DIFUSCO_WL_NORM_E_B),
                                               LW(l, DIFUSCO_WL_OUT_LN_W), LW(l, DIFUSCO_WL_OUT_LN_B),
                                               tbias + (size_t)l * H, tsp ? 1 : 0, st, a->aggregation))
    PROF(PROF_LINEAR_EDGE, edge_linear(ws.tmp, LW(l, DIFUSCO_WL_OUT_W), LW(l, DIFUSCO_WL_OUT_PLANES),
                                       LW(l, DIFUSCO_WL_OUT_B), ws.e, ws.e))
  }

  // DIFUSCO_FLAG_CHECK_FINITE: count inf / nan in the outputs (the statistics scratch is free after the head), synchronise
  auto finish = [&]() -> int {
    if ((a->flags & DIFUSCO_FLAG_CHECK_FINITE) == 0 || a->gn_phase == 1) return DIFUSCO_OK;
    unsigned* cnt = reinterpret_cast<unsigned*>(ws.partial);
    HIP_TRY(hipMemsetAsync(cnt, 0, sizeof(unsigned), st));
    HIP_TRY(launch_count_nonfinite(a->xt_out, out_rows, cnt, st));
    HIP_TRY(launch_count_nonfinite(ws.stats, (long long)a->n_segments * 64, cnt, st));   // head GroupNorm mean / rstd: nan as soon as the
                                                                                         // final state holds one (a sampled bit hides it)
    HIP_TRY(launch_count_nonfinite(a->pred_out, out_rows * C, cnt, st));
    HIP_TRY(launch_count_nonfinite(a->prob_out, a->diffusion == DIFUSCO_CATEGORICAL ? out_rows : 0, cnt, st));
    unsigned bad = 0;
    HIP_TRY(hipMemcpyAsync(&bad, cnt, sizeof(unsigned), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (bad) return fail(DIFUSCO_ENONFINITE, "denoise step produced %u non-finite output values", bad);
    return DIFUSCO_OK;
  };

  // head + posterior (gnn_encoder.py:400-401 / :412-413, pl_tsp_model.py:133-137, pl_meta_model.py:102-175)
  if (fused && tsp) {
    PROF(PROF_HEAD, launch_head_tiled(C, ws.e, E, gn_blocks_for(out_rows) < 8 ? 8 : gn_blocks_for(out_rows) / 8 * 8,
                                      ws.partial, ws.stats, G(DIFUSCO_W_OUT_GN_W), G(DIFUSCO_W_OUT_GN_B),
                                      G(DIFUSCO_W_OUT_CONV_W), G(DIFUSCO_W_OUT_CONV_B), a->perm, a->xt, a->post,
                                      a->rand_mode, a->rand, a->seed, a->offset, a->xt_out, a->pred_out, a->prob_out, st,
                                      gn_fold ? ws.gn_tile : nullptr, a->gn_phase, a->gn_sums,
                                      a->n_segments > 1 ? a->seg_ptr : nullptr, a->n_segments))
    return finish();
  }
  PROF(PROF_HEAD, launch_head(H, C, tsp ? ws.e : ws.h, a->n_segments > 1 ? a->seg_ptr : nullptr, a->n_segments, out_rows,
                              gn_blocks_for(out_rows), ws.partial, ws.stats, G(DIFUSCO_W_OUT_GN_W), G(DIFUSCO_W_OUT_GN_B),
                              G(DIFUSCO_W_OUT_CONV_W), G(DIFUSCO_W_OUT_CONV_B), tsp ? a->perm : nullptr, a->xt, a->post,
                              a->rand_mode, a->rand, a->seed, a->offset, a->xt_out, a->pred_out, a->prob_out, st,
                              a->gn_phase, a->gn_sums))
#undef PROF
  return finish();
}

size_t difusco_prepared_bytes(int hidden, int n_nodes) {
  if (!hidden_ok(hidden) || n_nodes <= 0) return 0;
  return carve_prepared(nullptr, hidden, n_nodes).bytes;
}

int difusco_time_bias_rows(int hidden, int n_layers, int out_channels, const float* weights, const float* t_host, int n_t,
                           float* out, void* stream) {
  using namespace difusco;
  if (!hidden_ok(hidden) || n_layers < 1 || (out_channels != 1 && out_channels != 2))
    return fail(DIFUSCO_EINVAL, "hidden in {64,128,256}, n_layers >= 1, out_channels in {1,2} required");
  if (!weights || !t_host || !out || n_t < 1) return fail(DIFUSCO_EINVAL, "null pointer or n_t < 1");
  const Layout lo = make_layout(hidden, n_layers, out_channels);
  auto G = [&](int id) { return weights + lo.off[id]; };
  HIP_TRY(launch_time_bias(t_host, n_t, hidden, n_layers, G(DIFUSCO_W_TIME_FREQS), G(DIFUSCO_W_TIME0_W), G(DIFUSCO_W_TIME0_B),
                           G(DIFUSCO_W_TIME2_W), G(DIFUSCO_W_TIME2_B), weights + lo.off[DIFUSCO_W_GLOBAL_COUNT], lo.layer_stride,
                           lo.off[DIFUSCO_W_GLOBAL_COUNT + DIFUSCO_WL_TIME_W] - lo.off[DIFUSCO_W_GLOBAL_COUNT],
                           lo.off[DIFUSCO_W_GLOBAL_COUNT + DIFUSCO_WL_TIME_B] - lo.off[DIFUSCO_W_GLOBAL_COUNT], out,
                           (hipStream_t)stream));
  return DIFUSCO_OK;
}

// e0 = edge_embed(ScalarEmbeddingSine(x_t)) for a general (Gaussian / non-binary) x_t, by the kernel the fused step uses (edge_embed.hip):
// exported for the parity tests (partial tiles, permuted inputs, |x_t| up to 6).
size_t difusco_gen_table_bytes(int hidden) {
  if (hidden != 256) return 0;
  // rows | build scratch: the sinusoidal features of the grid, the grid itself
  return sizeof(float) * ((size_t)2 * difusco::kGenRows * 256 + 1024);
}

int difusco_gen_table_build(int hidden, int n_layers, int out_channels, const float* weights, float* table, size_t table_bytes,
                            void* stream) {
  using namespace difusco;
  if (hidden != 256) return fail(DIFUSCO_EINVAL, "difusco_gen_table_build: hidden = 256 required");
  if (n_layers < 1 || (out_channels != 1 && out_channels != 2) || !weights || !table) return fail(DIFUSCO_EINVAL, "bad model shape / null pointer");
  if (table_bytes < difusco_gen_table_bytes(hidden)) return fail(DIFUSCO_EWORKSPACE, "gen table buffer too small: %zu < %zu", table_bytes, difusco_gen_table_bytes(hidden));
  const Layout lo = make_layout(hidden, n_layers, out_channels);
  auto G = [&](int id) { return weights + lo.off[id]; };
  const int H = hidden;
  hipStream_t st = (hipStream_t)stream;
  float* feat = table + (size_t)kGenRows * H;
  float* grid = feat + (size_t)kGenRows * H;
  HIP_TRY(launch_gen_table_grid(grid, st));
  HIP_TRY(launch_scalar_embed(grid, nullptr, G(DIFUSCO_W_DIMT_SCALAR), kGenRows, H, feat, st));       // precise sin / cos of x / dim_t
  HIP_TRY(linear_rows(feat, G(DIFUSCO_W_EDGE_EMBED_W), G(DIFUSCO_W_EDGE_EMBED_B), nullptr, table, kGenRows, H, H, H, st));   // exact fp32 MFMA
  return DIFUSCO_OK;
}

int difusco_edge_embed(int hidden, int n_layers, int out_channels, const float* weights, int precision, const float* xt,
                       const int32_t* perm, int64_t n_edges, float* e_tiled, float* tile_max, const float* gen_table, void* stream) {
  using namespace difusco;
  if (hidden != 256) return fail(DIFUSCO_EINVAL, "difusco_edge_embed: the tiled kernel exists for hidden = 256");
  if (n_layers < 1 || (out_channels != 1 && out_channels != 2)) return fail(DIFUSCO_EINVAL, "n_layers >= 1, out_channels in {1,2} required");
  if (precision != DIFUSCO_PREC_BF16X3 && precision != DIFUSCO_PREC_FP16X3)
    return fail(DIFUSCO_EINVAL, "difusco_edge_embed: precision must be BF16X3 or FP16X3");
  if (!weights || !xt || !e_tiled || n_edges < 0 || n_edges > 0x7fffffff) return fail(DIFUSCO_EINVAL, "null pointer / bad n_edges");
  const Layout lo = make_layout(hidden, n_layers, out_channels);
  auto G = [&](int id) { return weights + lo.off[id]; };
  const int H = hidden;
  const bool f16 = precision == DIFUSCO_PREC_FP16X3;
  const unsigned short* pl = reinterpret_cast<const unsigned short*>(G(DIFUSCO_W_EDGE_EMBED_PLANES)) + (f16 ? (long long)3 * H * H : 0);
  // (stand-alone entry: the tile flags live in the table buffer's build scratch - enough for 581,632 edges; the step uses its workspace)
  int* flags = gen_table ? reinterpret_cast<int*>(const_cast<float*>(gen_table) + (size_t)kGenRows * H) : nullptr;
  if (gen_table && (n_edges + 127) / 128 * 4 > (int64_t)kGenRows * H)
    return fail(DIFUSCO_EINVAL, "difusco_edge_embed with a table: at most %d edges per call (use the step for larger inputs)", kGenRows * H * 32);
  HIP_TRY(launch_edge_embed_tiled(xt, perm, G(DIFUSCO_W_DIMT_SCALAR), pl, (long long)H * H, precision,
                                  f16 ? G(DIFUSCO_W_EDGE_EMBED_PLANES) + (long long)5 * H * H / 2 : nullptr, G(DIFUSCO_W_EDGE_EMBED_B),
                                  e_tiled, n_edges, tile_max, (hipStream_t)stream, gen_table, flags));
  return DIFUSCO_OK;
}

// The step-invariant part of a TSP step, with the kernels (and therefore the bits) of the stateless step: node embedding,
// row scales, layer 0's node linear, the two-row edge-input table and C of layer 0 applied to it.
int difusco_prepare(const difusco_step_args* a, void* prepared, size_t prepared_bytes) {
  using namespace difusco;
  if (!a || !prepared) return fail(DIFUSCO_EINVAL, "null args / buffer");
  if (a->struct_size != sizeof(difusco_step_args) || a->abi_version != DIFUSCO_ABI_VERSION)
    return fail(DIFUSCO_EINVAL, "ABI mismatch: struct_size %u (want %zu), abi %u (want %d)", a->struct_size,
                sizeof(difusco_step_args), a->abi_version, DIFUSCO_ABI_VERSION);
  const int H = a->hidden, L = a->n_layers, C = a->out_channels;
  if (!hidden_ok(H) || L < 1 || (C != 1 && C != 2)) return fail(DIFUSCO_EINVAL, "bad model shape");
  if (a->task != DIFUSCO_TASK_TSP) return fail(DIFUSCO_EINVAL, "prepared state exists for TSP only (the MIS node embedding is a function of x_t)");
  const int64_t N = a->n_nodes, E = a->n_edges;
  if (N <= 0 || E < 0 || !a->weights || !a->points || !a->workspace) return fail(DIFUSCO_EINVAL, "null pointer / empty graph");
  if (a->precision < DIFUSCO_PREC_FP32 || a->precision > DIFUSCO_PREC_FP16X3) return fail(DIFUSCO_EINVAL, "unknown precision %d", a->precision);
  const Prepared prep = carve_prepared(prepared, H, N);
  if (prep.bytes > prepared_bytes) return fail(DIFUSCO_EWORKSPACE, "prepared buffer too small: %zu < %zu", prepared_bytes, prep.bytes);
  const int nblk = gn_blocks_for(E > N ? E : N);
  Workspace ws = carve(a->workspace, H, L, N, E, a->n_segments < 1 ? 1 : a->n_segments, nblk);
  if (ws.bytes > a->workspace_bytes) return fail(DIFUSCO_EWORKSPACE, "workspace too small: %zu < %zu", a->workspace_bytes, ws.bytes);
  const Layout lo = make_layout(H, L, C);
  const float* W = a->weights;
  auto G = [&](int id) { return W + lo.off[id]; };
  auto LW = [&](int l, int id) { return W + lo.off[DIFUSCO_W_GLOBAL_COUNT + l * DIFUSCO_WL_COUNT + id]; };
  hipStream_t st = (hipStream_t)a->stream;
  const bool f16 = a->precision == DIFUSCO_PREC_FP16X3;
  // h0 = node_embed(pos_embed(points))  (ws.node4 is the scratch of the sinusoidal features, as in the step)
  HIP_TRY(launch_pos_embed(a->points, G(DIFUSCO_W_DIMT_POS), (int)N, H, ws.node4, st));
  HIP_TRY(linear_rows(ws.node4, G(DIFUSCO_W_NODE_EMBED_W), G(DIFUSCO_W_NODE_EMBED_B), nullptr, prep.h0, N, H, H, H, st));
  // layer 0's U|V|A|B rows of h0
  if (a->precision != DIFUSCO_PREC_FP32 && H == 256) {
    const unsigned short* npl = reinterpret_cast<const unsigned short*>(LW(0, DIFUSCO_WL_NODE4_PLANES)) +
                                (f16 ? (long long)3 * 4 * H * H : 0);
    SplitScale sc;
    if (f16) {
      HIP_TRY(launch_row_pow2_scale(prep.h0, N, H, ws.hscale, st));
      sc.row_scale = ws.hscale;
      sc.w_inv = LW(0, DIFUSCO_WL_NODE4_PLANES) + (long long)5 * 4 * H * H / 2;
    }
    // the prepared rows are read by the fused edge kernel only (difusco_denoise_step: use_prep requires the fused path), so they
    // are produced in its log2(e) domain whenever the precision has a fused path - same vectors as the step's own node linear
    const bool fused_prec = a->precision == DIFUSCO_PREC_BF16X3 || a->precision == DIFUSCO_PREC_FP16X3;
    if (fused_prec) sc.w_inv = LW(0, DIFUSCO_WL_NODE4_FUSED_S) + (f16 ? 0 : 4 * H);
    HIP_TRY(linear_rows_split(prep.h0, npl, (long long)4 * H * H, a->precision,
                              LW(0, fused_prec ? DIFUSCO_WL_NODE4_FUSED_B : DIFUSCO_WL_NODE4_B), nullptr,
                              prep.node4_0, N, H, 4 * H, 4 * H, st, 0, sc));
    //  a precision without a fused path (BF16X6) computes the reference's rows: converted in place, so that the buffer
    // is in the fused kernel's domain WHATEVER precision prepared it - a step that runs a fused precision on it then differs by
    // the rounding of the node linear only (as under ABI 10), never by a missing b_C / log2(e)
    if (!fused_prec) HIP_TRY(launch_fuse_node_tables(prep.node4_0, LW(0, DIFUSCO_WL_C_B), (int)N, prep.node4_0, st));
  } else {
    HIP_TRY(linear_rows(prep.h0, LW(0, DIFUSCO_WL_NODE4_W), LW(0, DIFUSCO_WL_NODE4_B), nullptr, prep.node4_0, N, H, 4 * H,
                        4 * H, st));
    if (H == 256) HIP_TRY(launch_fuse_node_tables(prep.node4_0, LW(0, DIFUSCO_WL_C_B), (int)N, prep.node4_0, st));      // (FP32: same rule)
  }
  // two-row edge-input table of a categorical step (embedding of the bit) and C of layer 0 applied to it
  HIP_TRY(launch_scalar_embed(nullptr, nullptr, G(DIFUSCO_W_DIMT_SCALAR), 2, H, ws.table_in, st));
  HIP_TRY(launch_two_rows_linear(H, ws.table_in, G(DIFUSCO_W_EDGE_EMBED_W), G(DIFUSCO_W_EDGE_EMBED_B), prep.table, st));
  HIP_TRY(launch_two_rows_linear(H, prep.table, LW(0, DIFUSCO_WL_C_W), nullptr, prep.table + 2 * H, st));
  return DIFUSCO_OK;
}

int difusco_linear_rows(const float* x, const float* w, const float* bias, const float* residual, float* y, int64_t m,
                        int k, int n_out, int64_t ldy, void* stream) {
  if (!x || !w || !y) return fail(DIFUSCO_EINVAL, "null pointer");
  if (!(k == 32 || k == 64 || k == 128 || k == 256) || n_out % 32 != 0 || n_out <= 0 || ldy < n_out)
    return fail(DIFUSCO_EINVAL, "k must be 32/64/128/256, n_out a positive multiple of 32, ldy >= n_out");
  HIP_TRY(difusco::linear_rows(x, w, bias, residual, y, m, k, n_out, ldy, (hipStream_t)stream));
  return DIFUSCO_OK;
}

int difusco_linear_rows_split(const float* x, const void* planes, int precision, const float* bias,
                              const float* residual, float* y, int64_t m, int k, int n_out, int64_t ldy,
                              float* row_scale_scratch, void* stream) {
  if (!x || !planes || !y) return fail(DIFUSCO_EINVAL, "null pointer");
  if (precision < DIFUSCO_PREC_BF16X3 || precision > DIFUSCO_PREC_FP16X3)
    return fail(DIFUSCO_EINVAL, "precision must be BF16X3, BF16X6 or FP16X3");
  if (!(k == 64 || k == 128 || k == 256) || (n_out != k && !(k == 256 && n_out > 0 && n_out % 256 == 0)) || ldy < n_out)
    return fail(DIFUSCO_EINVAL, "split path needs k in {64,128,256}, n_out == k (or a multiple of 256 for k = 256), ldy >= n_out");
  const unsigned short* pl = reinterpret_cast<const unsigned short*>(planes);
  difusco::SplitScale sc;
  if (precision == DIFUSCO_PREC_FP16X3) {
    pl += (long long)3 * n_out * k;
    sc.w_inv = reinterpret_cast<const float*>(planes) + (long long)5 * n_out * k / 2;
    if (row_scale_scratch != nullptr) {
      HIP_TRY(difusco::launch_row_pow2_scale(x, m, k, row_scale_scratch, (hipStream_t)stream));
      sc.row_scale = row_scale_scratch;
    }
  }
  HIP_TRY(difusco::linear_rows_split(x, pl, (long long)n_out * k, precision, bias, residual, y, m, k, n_out, ldy,
                                     (hipStream_t)stream, 0, sc));
  return DIFUSCO_OK;
}

int difusco_edge_gate_aggregate(int hidden, int n_nodes, const int32_t* rowptr, const int32_t* col, const float* node4,
                                float* ce_act, float* h, const float* norm_h_w, const float* norm_h_b,
                                const float* norm_e_w, const float* norm_e_b, const float* out_ln_w,
                                const float* out_ln_b, const float* tbias, int time_on_edge, void* stream) {
  if (!hidden_ok(hidden)) return fail(DIFUSCO_EINVAL, "hidden must be 64, 128 or 256");
  if (!rowptr || !node4 || !h || !norm_h_w || !norm_h_b || !norm_e_w || !norm_e_b || !out_ln_w || !out_ln_b || !tbias)
    return fail(DIFUSCO_EINVAL, "null pointer");
  HIP_TRY(difusco::launch_edge_gate_aggregate(hidden, n_nodes, rowptr, col, node4, ce_act, h, norm_h_w, norm_h_b, norm_e_w,
                                              norm_e_b, out_ln_w, out_ln_b, tbias, time_on_edge, (hipStream_t)stream));
  return DIFUSCO_OK;
}

size_t difusco_fused_scratch_bytes(int n_nodes, int n_edges) {
  if (n_nodes < 0 || n_edges < 0) return 0;
  // pieces of the neighbour sum | direct rows | tile maxima | the node rows in the kernel's log2(e) domain (ABI 11)
  return sizeof(float) * (fused_part_floats(n_edges) + (size_t)n_nodes * 256 + (size_t)(n_edges + 255) / 256 * 8 + 64 + 64 +
                          (size_t)n_nodes * 1024) + 256;
}

int difusco_edge_layer_fused(int precision, int n_nodes, int n_edges, const int32_t* rowptr, const int32_t* row,
                             const int32_t* col, const float* node4, float* e, float* h, const void* planes_c,
                             const void* planes_o, const float* b_c, const float* norm_h_w, const float* norm_h_b,
                             const float* norm_e_w, const float* norm_e_b, const float* out_ln_w,
                             const float* out_ln_b, const float* b_out, const float* tbias, int time_on_edge,
                             const float* scales, void* scratch, void* stream) {
  if (precision != DIFUSCO_PREC_BF16X3 && precision != DIFUSCO_PREC_FP16X3)
    return fail(DIFUSCO_EINVAL, "fused kernel: precision must be BF16X3 or FP16X3");
  if (!rowptr || !row || !col || !node4 || !e || !h || !planes_c || !planes_o || !b_c || !norm_h_w || !norm_h_b ||
      !norm_e_w || !norm_e_b || !out_ln_w || !out_ln_b || !b_out || !tbias || !scratch)
    return fail(DIFUSCO_EINVAL, "null pointer");
  if (precision == DIFUSCO_PREC_FP16X3 && !scales)
    return fail(DIFUSCO_EINVAL, "fused kernel, FP16X3: the operand-scale record of the layer is required");
  const long long off = precision == DIFUSCO_PREC_FP16X3 ? 3LL * 256 * 256 : 0;
  float* part = reinterpret_cast<float*>(scratch);
  float* direct = part + (fused_part_floats(n_edges) + 63) / 64 * 64;
  float* etmax = direct + (size_t)n_nodes * 256;
  hipStream_t st = (hipStream_t)stream;
  const long long n_tiles_pad = ((long long)n_edges + 255) / 256 * 8;
  // The caller's node4 is the reference's U h | V h | A h | B h (gnn_encoder.py:94-103).  The kernel reads the A | B rows in its
  // log2(e) domain with b_C folded into the A rows (difusco_hip.h, ABI 11 note): converted copy in the scratch.  (The step driver
  // has no such pass: its node linear produces the rows in that form.)
  float* node4_k = etmax + (n_tiles_pad + 63) / 64 * 64;
  HIP_TRY(difusco::launch_fuse_node_tables(node4, b_c, n_nodes, node4_k, st));
  if (precision == DIFUSCO_PREC_FP16X3)      // e-stream scale per tile: normally left by the producer of e
    HIP_TRY(difusco::launch_tile_absmax_tiled(e, n_tiles_pad, etmax, st));
  HIP_TRY(difusco::launch_edge_layer_fused(precision, e, node4_k, row, col, n_edges,
                                           reinterpret_cast<const unsigned short*>(planes_c) + off,
                                           reinterpret_cast<const unsigned short*>(planes_o) + off, 256LL * 256, b_c,
                                           norm_e_w, norm_e_b, tbias, out_ln_w, out_ln_b, b_out, time_on_edge, part,
                                           direct, scales, etmax, etmax, st, n_nodes >= (1 << 20) ? 1 : 0));
  // (node_finalize reads the U rows, which the conversion copies unchanged: either buffer would do)
  HIP_TRY(difusco::launch_node_finalize(n_nodes, n_edges, rowptr, node4, part, direct, h, norm_h_w, norm_h_b, tbias,
                                        time_on_edge, nullptr, st));
  return DIFUSCO_OK;
}

int difusco_categorical_posterior(const float* logits, const float* xt, const float* post, int rand_mode,
                                  const float* rand, uint64_t seed, uint64_t offset, float* xt_out, float* prob_out,
                                  int64_t n, void* stream) {
  if (!logits || !xt || !post || !xt_out) return fail(DIFUSCO_EINVAL, "null pointer");
  if (post[4] != 0.0f && (rand_mode == DIFUSCO_RAND_NONE || (rand_mode == DIFUSCO_RAND_INJECTED && !rand)))
    return fail(DIFUSCO_EINVAL, "this step draws random numbers: provide rand or use PHILOX");
  HIP_TRY(difusco::launch_categorical_posterior(logits, xt, post, rand_mode, rand, seed, offset, xt_out, prob_out, n,
                                                (hipStream_t)stream));
  return DIFUSCO_OK;
}

int difusco_gaussian_posterior(const float* pred, const float* xt, const float* post, int rand_mode, const float* rand,
                               uint64_t seed, uint64_t offset, float* xt_out, int64_t n, void* stream) {
  if (!pred || !xt || !post || !xt_out) return fail(DIFUSCO_EINVAL, "null pointer");
  if (post[4] != 0.0f && (rand_mode == DIFUSCO_RAND_NONE || (rand_mode == DIFUSCO_RAND_INJECTED && !rand)))
    return fail(DIFUSCO_EINVAL, "this step draws random numbers: provide rand or use PHILOX");
  HIP_TRY(difusco::launch_gaussian_posterior(pred, xt, post, rand_mode, rand, seed, offset, xt_out, n, (hipStream_t)stream));
  return DIFUSCO_OK;
}

#ifdef DIFUSCO_PROFILING
int difusco_debug_set_ptr(int key, void* p) {
  if (key == 1) { difusco::g_fused_dbg = reinterpret_cast<unsigned long long*>(p); return DIFUSCO_OK; }
  return fail(DIFUSCO_EINVAL, "unknown debug key %d", key);
}

int difusco_debug_set(int key, int value) {
  if (key == 0) { difusco::g_fused_ablate = value; return DIFUSCO_OK; }
  if (key == 6) { difusco::g_fused_lds_pad = value; return DIFUSCO_OK; }
  if (key == 9) { difusco::g_fused_start_delay = value; return DIFUSCO_OK; }
  if (key == 7) { difusco::g_fused_opt = value; return DIFUSCO_OK; }
  if (key == 10 && value >= 0 && value <= 31) { difusco::g_node_linear_ablate = value; return DIFUSCO_OK; }
  if (key == 11 && value >= 0 && value <= 3) { g_step_skip = value; return DIFUSCO_OK; }      // bit 0: no node_finalize launches, bit 1: no node linears from layer 2 on
  if (key == 8 && (value == 0 || value == 1 || value == 4)) { difusco::g_node_linear_depth = value; return DIFUSCO_OK; }
  return fail(DIFUSCO_EINVAL, "unknown debug key %d", key);
}
#endif  // DIFUSCO_PROFILING

int difusco_profile_enable(int on, int max_launches) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (on) {
    if (max_launches < 1) max_launches = 1;
    while (g_prof.ev.size() < (size_t)max_launches * 2) {
      hipEvent_t e;
      HIP_TRY(hipEventCreate(&e));
      g_prof.ev.push_back(e);
    }
    g_prof.cat.assign(g_prof.ev.size() / 2, 0);
    g_prof.used = 0;
  }
  g_prof.on = on != 0;
  g_prof.dominant_only = on == 2;   // 2: only the dominant category (0) is bracketed - the cheap mode bench.py uses
  return DIFUSCO_OK;
}

int difusco_profile_collect(double* ms, int64_t* launches, int n_categories) {
  if (!ms || !launches || n_categories < PROF_NCAT) return fail(DIFUSCO_EINVAL, "need %d categories", PROF_NCAT);
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (int c = 0; c < n_categories; ++c) { ms[c] = 0.0; launches[c] = 0; }
  for (size_t i = 0; i < g_prof.used; ++i) {
    HIP_TRY(hipEventSynchronize(g_prof.ev[2 * i + 1]));
    float t = 0.0f;
    HIP_TRY(hipEventElapsedTime(&t, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]));
    ms[g_prof.cat[i]] += t;
    launches[g_prof.cat[i]] += 1;
  }
  const int used = (int)g_prof.used;
  g_prof.used = 0;
  return used;
}

}  // extern "C"
