// pointnet2_aten.cpp -- an ATen / pybind11 front end of the C ABI (include/butd_pointnet2.h) exporting the
// nine functions of the reference's native module `pointnet2._ext`
// (nickgkan/butd_detr pointnet2/_ext_src/src/bindings.cpp:11-24) with its calling conventions:
//   * tensors must be contiguous, fp32 data / int32 indices, all on the GPU of the first argument
//     (the reference's CHECK_* macros, include/utils.h:10-30);  CPU tensors -> "CPU not supported";
//   * outputs are allocated here on the input's device (sampling.cpp:30-32,74-80 ...);
//   * launches go to torch's CURRENT stream of that device, asynchronously.
// This is the binding a maintainer keeps if the reference's own `_ext` build is to stay in place: the
// `*_kernel_wrapper` prototypes of src/{sampling,ball_query,group_points,interpolate}.cpp become calls of the
// butd_* entry points.  Built by butd_detr_amd/binding/build.py (torch.utils.cpp_extension, in-tree); the
// product path itself binds the same C ABI with ctypes (butd_detr_amd/_hiplib.py) and does not need it.
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <c10/hip/HIPStream.h>
#include <torch/extension.h>

#include <vector>

#include "butd_pointnet2.h"

namespace {

// layout and type of every argument first, devices afterwards -- the order of the reference's shims
// (CHECK_CONTIGUOUS / CHECK_IS_FLOAT / CHECK_IS_INT, then the is_cuda() branch)
struct Arg {
  const at::Tensor &t;
  at::ScalarType dtype;
  const char *name;
};
// ... and make that GPU the current device for the rest of the entry point: output allocations, the workspace
// queries and the launches inside the C ABI all act on the CURRENT device, which need not be the tensors' one
using DeviceGuard = c10::hip::HIPGuardMasqueradingAsCUDA;
DeviceGuard need(std::initializer_list<Arg> args) {
  for (const Arg &a : args) {
    TORCH_CHECK(a.t.is_contiguous(), a.name, " must be a contiguous tensor");
    TORCH_CHECK(a.t.scalar_type() == a.dtype, a.name, " must be a ", a.dtype == at::kFloat ? "float" : "int", " tensor");
  }
  const at::Tensor &anchor = args.begin()->t;
  TORCH_CHECK(anchor.is_cuda(), "CPU not supported");
  for (const Arg &a : args)
    TORCH_CHECK(a.t.is_cuda() && a.t.device() == anchor.device(), a.name, " must be a CUDA tensor on ", anchor.device());
  return DeviceGuard(anchor.device());
}

butd_stream_t stream_of(const at::Tensor &t) {
  return (butd_stream_t)c10::hip::getCurrentHIPStream(t.device().index()).stream();
}

void ok(int err, const char *what) { TORCH_CHECK(err == 0, what, ": ", butd_error_string(err)); }

at::TensorOptions like(const at::Tensor &t, at::ScalarType dtype) { return at::device(t.device()).dtype(dtype); }

at::Tensor furthest_point_sampling(at::Tensor points, const int nsamples) {   // sampling.cpp:70-91
  DeviceGuard on_device = need({{points, at::kFloat, "points"}});
  const int b = points.size(0), n = points.size(1);
  at::Tensor idx = torch::zeros({b, nsamples}, like(points, at::kInt));
  at::Tensor temp = torch::empty({b, n}, like(points, at::kFloat));
  const size_t ws_bytes = butd_fps_workspace_bytes(b, n);
  if (ws_bytes) {   // exact spatially pruned path for large clouds (same indices)
    at::Tensor ws = torch::empty({(long)ws_bytes}, like(points, at::kByte));
    ok(butd_furthest_point_sampling_ws(b, n, nsamples, points.data_ptr<float>(), temp.data_ptr<float>(),
                                       idx.data_ptr<int>(), ws.data_ptr(), ws_bytes, stream_of(points)),
       "furthest_point_sampling");
  } else {
    ok(butd_furthest_point_sampling(b, n, nsamples, points.data_ptr<float>(), temp.data_ptr<float>(),
                                    idx.data_ptr<int>(), stream_of(points)),
       "furthest_point_sampling");
  }
  return idx;
}

at::Tensor gather_points(at::Tensor points, at::Tensor idx) {   // sampling.cpp:20-45
  DeviceGuard on_device = need({{points, at::kFloat, "points"}, {idx, at::kInt, "idx"}});
  at::Tensor out = torch::empty({points.size(0), points.size(1), idx.size(1)}, like(points, at::kFloat));
  ok(butd_gather_points(points.size(0), points.size(1), points.size(2), idx.size(1), points.data_ptr<float>(),
                        idx.data_ptr<int>(), out.data_ptr<float>(), stream_of(points)),
     "gather_points");
  return out;
}

at::Tensor gather_points_grad(at::Tensor grad_out, at::Tensor idx, const int n) {   // sampling.cpp:47-69
  DeviceGuard on_device = need({{grad_out, at::kFloat, "grad_out"}, {idx, at::kInt, "idx"}});
  at::Tensor out = torch::zeros({grad_out.size(0), grad_out.size(1), n}, like(grad_out, at::kFloat));
  ok(butd_gather_points_grad(grad_out.size(0), grad_out.size(1), n, idx.size(1), grad_out.data_ptr<float>(),
                             idx.data_ptr<int>(), out.data_ptr<float>(), stream_of(grad_out)),
     "gather_points_grad");
  return out;
}

// note the argument order: centres first (include/ball_query.h:9-10)
at::Tensor ball_query(at::Tensor new_xyz, at::Tensor xyz, const float radius, const int nsample) {
  DeviceGuard on_device = need({{new_xyz, at::kFloat, "new_xyz"}, {xyz, at::kFloat, "xyz"}});
  const int b = xyz.size(0), n = xyz.size(1), m = new_xyz.size(1);
  at::Tensor idx = torch::empty({b, m, nsample}, like(new_xyz, at::kInt));
  const size_t ws_bytes = butd_ball_query_workspace_bytes(b, n, m);
  at::Tensor ws = torch::empty({(long)ws_bytes}, like(new_xyz, at::kByte));
  ok(butd_ball_query_ws(b, n, m, radius, nsample, new_xyz.data_ptr<float>(), xyz.data_ptr<float>(),
                        idx.data_ptr<int>(), ws_bytes ? ws.data_ptr() : nullptr, ws_bytes, stream_of(new_xyz)),
     "ball_query");
  return idx;
}

at::Tensor group_points(at::Tensor points, at::Tensor idx) {   // group_points.cpp:17-39
  DeviceGuard on_device = need({{points, at::kFloat, "points"}, {idx, at::kInt, "idx"}});
  at::Tensor out = torch::empty({points.size(0), points.size(1), idx.size(1), idx.size(2)}, like(points, at::kFloat));
  ok(butd_group_points(points.size(0), points.size(1), points.size(2), idx.size(1), idx.size(2),
                       points.data_ptr<float>(), idx.data_ptr<int>(), out.data_ptr<float>(), stream_of(points)),
     "group_points");
  return out;
}

at::Tensor group_points_grad(at::Tensor grad_out, at::Tensor idx, const int n) {   // group_points.cpp:41-65
  DeviceGuard on_device = need({{grad_out, at::kFloat, "grad_out"}, {idx, at::kInt, "idx"}});
  at::Tensor out = torch::zeros({grad_out.size(0), grad_out.size(1), n}, like(grad_out, at::kFloat));
  ok(butd_group_points_grad(grad_out.size(0), grad_out.size(1), n, idx.size(1), idx.size(2),
                            grad_out.data_ptr<float>(), idx.data_ptr<int>(), out.data_ptr<float>(),
                            stream_of(grad_out)),
     "group_points_grad");
  return out;
}

std::vector<at::Tensor> three_nn(at::Tensor unknowns, at::Tensor knows) {   // interpolate.cpp:19-46
  DeviceGuard on_device = need({{unknowns, at::kFloat, "unknowns"}, {knows, at::kFloat, "knows"}});
  at::Tensor idx = torch::zeros({unknowns.size(0), unknowns.size(1), 3}, like(unknowns, at::kInt));
  at::Tensor dist2 = torch::zeros({unknowns.size(0), unknowns.size(1), 3}, like(unknowns, at::kFloat));
  ok(butd_three_nn(unknowns.size(0), unknowns.size(1), knows.size(1), unknowns.data_ptr<float>(),
                   knows.data_ptr<float>(), dist2.data_ptr<float>(), idx.data_ptr<int>(), stream_of(unknowns)),
     "three_nn");
  return {dist2, idx};
}

at::Tensor three_interpolate(at::Tensor points, at::Tensor idx, at::Tensor weight) {   // interpolate.cpp:48-76
  DeviceGuard on_device = need({{points, at::kFloat, "points"}, {idx, at::kInt, "idx"}, {weight, at::kFloat, "weight"}});
  at::Tensor out = torch::empty({points.size(0), points.size(1), idx.size(1)}, like(points, at::kFloat));
  ok(butd_three_interpolate(points.size(0), points.size(1), points.size(2), idx.size(1), points.data_ptr<float>(),
                            idx.data_ptr<int>(), weight.data_ptr<float>(), out.data_ptr<float>(), stream_of(points)),
     "three_interpolate");
  return out;
}

at::Tensor three_interpolate_grad(at::Tensor grad_out, at::Tensor idx, at::Tensor weight, const int m) {
  DeviceGuard on_device = need({{grad_out, at::kFloat, "grad_out"}, {idx, at::kInt, "idx"}, {weight, at::kFloat, "weight"}});   // interpolate.cpp:78-104
  at::Tensor out = torch::zeros({grad_out.size(0), grad_out.size(1), m}, like(grad_out, at::kFloat));
  ok(butd_three_interpolate_grad(grad_out.size(0), grad_out.size(1), grad_out.size(2), m,
                                 grad_out.data_ptr<float>(), idx.data_ptr<int>(), weight.data_ptr<float>(),
                                 out.data_ptr<float>(), stream_of(grad_out)),
     "three_interpolate_grad");
  return out;
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.def("gather_points", &gather_points);
  m.def("gather_points_grad", &gather_points_grad);
  m.def("furthest_point_sampling", &furthest_point_sampling);
  m.def("three_nn", &three_nn);
  m.def("three_interpolate", &three_interpolate);
  m.def("three_interpolate_grad", &three_interpolate_grad);
  m.def("ball_query", &ball_query);
  m.def("group_points", &group_points);
  m.def("group_points_grad", &group_points_grad);
}
