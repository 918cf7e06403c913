// PyTorch custom-op registration of the denoise step (BASELINE.json north_star: "exposed to the repo's Python through
// PyTorch-ROCm custom ops"; SURVEY.md section 8(b), last row).  A thin shim over the C ABI of libdifusco_hip.so
// (include/difusco_hip.h stays the primary boundary): tensors in, tensors out, the launch goes to the current HIP
// stream of the tensors' device.  Registered as torch.ops.difusco.*:
//
//   prepare_graph(edge_index, n_nodes) -> (rowptr, col, row, perm, identity)         host, CPU tensors
//   weights_layout(hidden, n_layers, out_channels) -> int64[entries + 1]              offsets, last = total floats
//   workspace_bytes(hidden, n_layers, n_nodes, n_edges, n_segments) -> int
//   prepare_state(weights, points, n_nodes, n_edges, n_segments, workspace, cfg) -> uint8 buffer     difusco_prepare
//   time_bias_rows(weights, times, cfg) -> float32 [n_t, n_layers, hidden]                            difusco_time_bias_rows
//   denoise_step_categorical(...) / denoise_step_gaussian(...) -> (xt_next, pred, prob)
//       replace {categorical,gaussian}_denoise_step of difusco/pl_tsp_model.py:122-151 / pl_mis_model.py:118-140
//
// Built by difusco_amd/build.py into difusco_amd/lib/libdifusco_torch.so (host compiler, links libdifusco_hip.so).
#include <ATen/hip/HIPContext.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <torch/library.h>
#include <torch/types.h>

#include <string>
#include <tuple>
#include <vector>

#include "../../include/difusco_hip.h"

namespace {

void check(int code, const char* what) {
  TORCH_CHECK(code >= 0, "libdifusco_hip ", what, " failed (", code, "): ", difusco_last_error());
}

const void* ptr_or_null(const c10::optional<at::Tensor>& t) {
  return (t.has_value() && t->defined()) ? t->data_ptr() : nullptr;
}

void need(const at::Tensor& t, at::ScalarType dt, const char* name, bool cuda) {
  TORCH_CHECK(t.scalar_type() == dt, name, ": wrong dtype");
  TORCH_CHECK(t.is_contiguous(), name, " must be contiguous");
  TORCH_CHECK(t.is_cuda() == cuda, name, cuda ? " must live on the GPU" : " must be a CPU tensor");
}

std::tuple<at::Tensor, at::Tensor, at::Tensor, at::Tensor, bool> prepare_graph(const at::Tensor& edge_index, int64_t n_nodes) {
  need(edge_index, at::kLong, "edge_index", false);
  TORCH_CHECK(edge_index.dim() == 2 && edge_index.size(0) == 2, "edge_index must be [2, E]");
  const int64_t E = edge_index.size(1);
  auto opt = at::TensorOptions().dtype(at::kInt);
  at::Tensor rowptr = at::empty({n_nodes + 1}, opt), col = at::empty({E}, opt), row = at::empty({E}, opt),
             perm = at::empty({E}, opt);
  int identity = 0;
  check(difusco_csr_from_coo_host(edge_index.data_ptr<int64_t>(), E, n_nodes, rowptr.data_ptr<int32_t>(),
                                  col.data_ptr<int32_t>(), row.data_ptr<int32_t>(), perm.data_ptr<int32_t>(), &identity),
        "difusco_csr_from_coo_host");
  return {rowptr, col, row, perm, identity != 0};
}

at::Tensor weights_layout(int64_t hidden, int64_t n_layers, int64_t out_channels) {
  const int n = DIFUSCO_W_GLOBAL_COUNT + (int)n_layers * DIFUSCO_WL_COUNT;
  at::Tensor out = at::empty({n + 1}, at::TensorOptions().dtype(at::kLong));
  int64_t total = 0;
  check(difusco_weights_layout((int)hidden, (int)n_layers, (int)out_channels, out.data_ptr<int64_t>(), n, &total),
        "difusco_weights_layout");
  out.data_ptr<int64_t>()[n] = total;
  return out;
}

int64_t workspace_bytes(int64_t hidden, int64_t n_layers, int64_t n_nodes, int64_t n_edges, int64_t n_segments) {
  return (int64_t)difusco_workspace_bytes((int)hidden, (int)n_layers, (int)n_nodes, (int)n_edges, (int)n_segments);
}

// One reverse-diffusion step on the current stream of `weights`' device.  `post` holds the 5 (categorical) / 5 (gaussian)
// host-computed posterior constants (include/difusco_hip.h: difusco_step_args.post); `cfg` = {hidden, n_layers,
// out_channels, task, precision, no_fusion, xt_is_binary, gn_phase, flags[, aggregation]}.
std::tuple<at::Tensor, at::Tensor, at::Tensor> step_impl(
    int diffusion, const at::Tensor& weights, const at::Tensor& rowptr, const at::Tensor& col,
    const c10::optional<at::Tensor>& perm, const c10::optional<at::Tensor>& row, const c10::optional<at::Tensor>& seg_ptr,
    const c10::optional<at::Tensor>& points, const at::Tensor& xt, double t, c10::ArrayRef<double> post,
    const c10::optional<at::Tensor>& rand, int64_t seed, int64_t offset, at::Tensor workspace, c10::ArrayRef<int64_t> cfg,
    bool want_pred, bool want_prob, const c10::optional<at::Tensor>& gn_sums, const c10::optional<at::Tensor>& prepared,
    const c10::optional<at::Tensor>& tbias, const c10::optional<at::Tensor>& gen_table) {
  TORCH_CHECK(cfg.size() == 9 || cfg.size() == 10,
              "cfg = {hidden, n_layers, out_channels, task, precision, no_fusion, xt_is_binary, gn_phase, flags[, aggregation]}");
  TORCH_CHECK(post.size() <= 8, "post holds at most 8 constants");
  need(weights, at::kFloat, "weights", true);
  need(rowptr, at::kInt, "rowptr", true);
  need(col, at::kInt, "col", true);
  need(xt, at::kFloat, "xt", true);
  TORCH_CHECK(workspace.is_cuda() && workspace.is_contiguous(), "workspace must be a contiguous GPU tensor");
  const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(weights.device());
  const int task = (int)cfg[3];
  const int64_t n_nodes = rowptr.numel() - 1, n_edges = col.numel();
  const int64_t rows = task == DIFUSCO_TASK_TSP ? n_edges : n_nodes;
  TORCH_CHECK(xt.numel() == rows, "xt has ", xt.numel(), " elements, the graph has ", rows, " output rows");
  const int C = (int)cfg[2];
  auto fo = xt.options();
  at::Tensor xt_out = at::empty({rows}, fo);
  at::Tensor pred = want_pred ? (C == 2 ? at::empty({rows, 2}, fo) : at::empty({rows}, fo)) : at::empty({0}, fo);
  at::Tensor prob = (want_prob && C == 2) ? at::empty({rows}, fo) : at::empty({0}, fo);

  difusco_step_args a{};
  a.struct_size = sizeof(difusco_step_args);
  a.abi_version = DIFUSCO_ABI_VERSION;
  a.hidden = (int)cfg[0];
  a.n_layers = (int)cfg[1];
  a.out_channels = C;
  a.task = task;
  a.weights = weights.data_ptr<float>();
  a.n_nodes = (int32_t)n_nodes;
  a.n_edges = (int32_t)n_edges;
  a.rowptr = rowptr.data_ptr<int32_t>();
  a.col = col.data_ptr<int32_t>();
  a.perm = static_cast<const int32_t*>(ptr_or_null(perm));
  a.row = static_cast<const int32_t*>(ptr_or_null(row));
  a.seg_ptr = static_cast<const int32_t*>(ptr_or_null(seg_ptr));
  a.n_segments = a.seg_ptr ? (int32_t)(seg_ptr->numel() - 1) : 1;
  a.points = static_cast<const float*>(ptr_or_null(points));
  a.xt = xt.data_ptr<float>();
  a.t = (float)t;
  a.xt_is_binary = (int32_t)cfg[6];
  a.diffusion = diffusion;
  for (size_t i = 0; i < post.size(); ++i) a.post[i] = (float)post[i];
  a.rand = static_cast<const float*>(ptr_or_null(rand));
  a.rand_mode = a.rand ? DIFUSCO_RAND_INJECTED : (a.post[4] != 0.0f ? DIFUSCO_RAND_PHILOX : DIFUSCO_RAND_NONE);
  a.seed = (uint64_t)seed;
  a.offset = (uint64_t)offset;
  a.xt_out = xt_out.data_ptr<float>();
  a.pred_out = want_pred ? pred.data_ptr<float>() : nullptr;
  a.prob_out = (want_prob && C == 2) ? prob.data_ptr<float>() : nullptr;
  a.workspace = workspace.data_ptr();
  a.workspace_bytes = (size_t)workspace.nbytes();
  a.stream = (void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(weights.device().index()).stream();
  a.precision = (int32_t)cfg[4];
  a.no_fusion = (int32_t)cfg[5];
  a.gn_phase = (int32_t)cfg[7];
  a.flags = (int32_t)cfg[8];
  a.gn_sums = static_cast<double*>(const_cast<void*>(ptr_or_null(gn_sums)));
  a.aggregation = cfg.size() > 9 ? (int32_t)cfg[9] : DIFUSCO_AGG_SUM;      // (ABI 10; a 9-entry cfg means "sum")
  a.prepared = ptr_or_null(prepared);                       // optional prepared state (include/difusco_hip.h, ABI 9)
  a.tbias = static_cast<const float*>(ptr_or_null(tbias));
  if (a.tbias) {
    need(*tbias, at::kFloat, "tbias", true);
    TORCH_CHECK(tbias->numel() == cfg[0] * cfg[1], "tbias must be [n_layers, hidden]");
  }
  a.gen_table = static_cast<const float*>(ptr_or_null(gen_table));      // optional generated-input table (ABI 12)
  if (a.gen_table) {
    need(*gen_table, at::kFloat, "gen_table", true);
    TORCH_CHECK((size_t)gen_table->nbytes() >= difusco_gen_table_bytes((int)cfg[0]) && difusco_gen_table_bytes((int)cfg[0]) > 0,
                "gen_table: the buffer of gen_table_build() required");
  }
  if (a.prepared)
    TORCH_CHECK(prepared->is_cuda() && prepared->is_contiguous() &&
                    (size_t)prepared->nbytes() >= difusco_prepared_bytes((int)cfg[0], (int)n_nodes),
                "prepared: contiguous GPU buffer of difusco_prepared_bytes() required");
  check(difusco_denoise_step(&a), "difusco_denoise_step");
  return {xt_out, pred, prob};
}

// Prepared state (include/difusco_hip.h, ABI 9): the step-invariant part of a TSP step for (weights, graph, points), and the
// time-bias rows of a whole schedule.  cfg as in step_impl (only hidden, n_layers, out_channels, precision are read).
at::Tensor prepare_state(const at::Tensor& weights, const at::Tensor& points, int64_t n_nodes, int64_t n_edges,
                         int64_t n_segments, at::Tensor workspace, c10::ArrayRef<int64_t> cfg) {
  TORCH_CHECK(cfg.size() == 9 || cfg.size() == 10,
              "cfg = {hidden, n_layers, out_channels, task, precision, no_fusion, xt_is_binary, gn_phase, flags[, aggregation]}");
  need(weights, at::kFloat, "weights", true);
  need(points, at::kFloat, "points", true);
  TORCH_CHECK(points.numel() == 2 * n_nodes, "points must be [n_nodes, 2]");
  TORCH_CHECK(workspace.is_cuda() && workspace.is_contiguous(), "workspace must be a contiguous GPU tensor");
  const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(weights.device());
  const size_t bytes = difusco_prepared_bytes((int)cfg[0], (int)n_nodes);
  TORCH_CHECK(bytes > 0, "difusco_prepared_bytes rejected the shape");
  at::Tensor out = at::empty({(int64_t)bytes}, weights.options().dtype(at::kByte));
  difusco_step_args a{};
  a.struct_size = sizeof(difusco_step_args);
  a.abi_version = DIFUSCO_ABI_VERSION;
  a.hidden = (int)cfg[0];
  a.n_layers = (int)cfg[1];
  a.out_channels = (int)cfg[2];
  a.task = DIFUSCO_TASK_TSP;
  a.weights = weights.data_ptr<float>();
  a.n_nodes = (int32_t)n_nodes;
  a.n_edges = (int32_t)n_edges;
  a.n_segments = (int32_t)n_segments;
  a.points = points.data_ptr<float>();
  a.workspace = workspace.data_ptr();
  a.workspace_bytes = (size_t)workspace.nbytes();
  a.stream = (void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(weights.device().index()).stream();
  a.precision = (int32_t)cfg[4];
  check(difusco_prepare(&a, out.data_ptr(), bytes), "difusco_prepare");
  return out;
}

at::Tensor time_bias_rows(const at::Tensor& weights, c10::ArrayRef<double> times, c10::ArrayRef<int64_t> cfg) {
  TORCH_CHECK((cfg.size() == 9 || cfg.size() == 10) && !times.empty(), "cfg (9 or 10 entries) and at least one time required");
  need(weights, at::kFloat, "weights", true);
  const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(weights.device());
  std::vector<float> t(times.begin(), times.end());
  at::Tensor out = at::empty({(int64_t)t.size(), cfg[1], cfg[0]}, weights.options());
  check(difusco_time_bias_rows((int)cfg[0], (int)cfg[1], (int)cfg[2], weights.data_ptr<float>(), t.data(), (int)t.size(),
                               out.data_ptr<float>(),
                               (void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(weights.device().index()).stream()),
        "difusco_time_bias_rows");
  return out;
}

// The generated-input table of a weight blob (include/difusco_hip.h, ABI 12): float32 buffer of difusco_gen_table_bytes().
at::Tensor gen_table_build(const at::Tensor& weights, c10::ArrayRef<int64_t> cfg) {
  TORCH_CHECK(cfg.size() >= 3, "cfg = {hidden, n_layers, out_channels, ...}");
  need(weights, at::kFloat, "weights", true);
  const c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(weights.device());
  const size_t bytes = difusco_gen_table_bytes((int)cfg[0]);
  TORCH_CHECK(bytes > 0, "difusco_gen_table_bytes rejected the shape (hidden = 256 required)");
  at::Tensor out = at::empty({(int64_t)(bytes / sizeof(float))}, weights.options());
  check(difusco_gen_table_build((int)cfg[0], (int)cfg[1], (int)cfg[2], weights.data_ptr<float>(), out.data_ptr<float>(), bytes,
                                (void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(weights.device().index()).stream()),
        "difusco_gen_table_build");
  return out;
}

#define STEP_SIGNATURE                                                                                                   \
  const at::Tensor &weights, const at::Tensor &rowptr, const at::Tensor &col, const c10::optional<at::Tensor>&perm,     \
      const c10::optional<at::Tensor>&row, const c10::optional<at::Tensor>&seg_ptr,                                     \
      const c10::optional<at::Tensor>&points, const at::Tensor &xt, double t, c10::ArrayRef<double> post,               \
      const c10::optional<at::Tensor>&rand, int64_t seed, int64_t offset, at::Tensor workspace,                         \
      c10::ArrayRef<int64_t> cfg, bool want_pred, bool want_prob, const c10::optional<at::Tensor>&gn_sums,             \
      const c10::optional<at::Tensor>&prepared, const c10::optional<at::Tensor>&tbias,                                  \
      const c10::optional<at::Tensor>&gen_table
#define STEP_FORWARD                                                                                                     \
  weights, rowptr, col, perm, row, seg_ptr, points, xt, t, post, rand, seed, offset, workspace, cfg, want_pred, want_prob, \
      gn_sums, prepared, tbias, gen_table

std::tuple<at::Tensor, at::Tensor, at::Tensor> denoise_step_categorical(STEP_SIGNATURE) {
  return step_impl(DIFUSCO_CATEGORICAL, STEP_FORWARD);
}
std::tuple<at::Tensor, at::Tensor, at::Tensor> denoise_step_gaussian(STEP_SIGNATURE) {
  return step_impl(DIFUSCO_GAUSSIAN, STEP_FORWARD);
}

const char* kStepSchema =
    "(Tensor weights, Tensor rowptr, Tensor col, Tensor? perm, Tensor? row, Tensor? seg_ptr, Tensor? points, Tensor xt, "
    "float t, float[] post, Tensor? rand, int seed, int offset, Tensor(a!) workspace, int[] cfg, bool want_pred, "
    "bool want_prob, Tensor(b!)? gn_sums, Tensor? prepared=None, Tensor? tbias=None, Tensor? gen_table=None) -> (Tensor, Tensor, Tensor)";

}  // namespace

TORCH_LIBRARY(difusco, m) {
  m.def("prepare_graph(Tensor edge_index, int n_nodes) -> (Tensor, Tensor, Tensor, Tensor, bool)");
  m.def("weights_layout(int hidden, int n_layers, int out_channels) -> Tensor", &weights_layout);
  m.def("workspace_bytes(int hidden, int n_layers, int n_nodes, int n_edges, int n_segments) -> int", &workspace_bytes);
  m.def("abi_version() -> int", []() -> int64_t { return difusco_abi_version(); });
  m.def("prepare_state(Tensor weights, Tensor points, int n_nodes, int n_edges, int n_segments, Tensor(a!) workspace, int[] cfg) -> Tensor");
  m.def("time_bias_rows(Tensor weights, float[] times, int[] cfg) -> Tensor");
  m.def("gen_table_build(Tensor weights, int[] cfg) -> Tensor");
  m.def((std::string("denoise_step_categorical") + kStepSchema).c_str());
  m.def((std::string("denoise_step_gaussian") + kStepSchema).c_str());
}

TORCH_LIBRARY_IMPL(difusco, CPU, m) { m.impl("prepare_graph", &prepare_graph); }

// ROCm builds of PyTorch dispatch HIP tensors under the CUDA key
TORCH_LIBRARY_IMPL(difusco, CUDA, m) {
  m.impl("prepare_state", &prepare_state);
  m.impl("time_bias_rows", &time_bias_rows);
  m.impl("gen_table_build", &gen_table_build);
  m.impl("denoise_step_categorical", &denoise_step_categorical);
  m.impl("denoise_step_gaussian", &denoise_step_gaussian);
}
