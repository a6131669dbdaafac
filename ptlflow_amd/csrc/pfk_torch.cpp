// torch.ops.pfk.* — the thin PyTorch-ROCm face of libpfk.so.
//
// The only translation unit that sees torch headers.  Each op checks device/dtype/layout,
// unwraps tensors to raw pointers + strides, picks up torch's *current* HIP stream and forwards
// to the C ABI of include/pfk.h; a non-zero status becomes a RuntimeError (the convention of
// the reference's own extension, ptlflow/utils/external/alt_cuda_corr/correlation.cpp:19-21).
// No kernel code, no allocation on the hot path (callers pass output/workspace tensors).
//
// Pixel-major activations are passed as 2-D tensors [M, C] whose row stride is the buffer's
// `ld` — a channel slice of a wider buffer (buf[:, 128:384]) is just a view.
#include <ATen/ATen.h>
#include <c10/core/DeviceGuard.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>

#include <vector>

#include "pfk.h"

namespace {

using at::Tensor;

void check_ok(int status, const char* what) {
  TORCH_CHECK(status == PFK_OK, "pfk::", what, " failed: ", pfk_status_string(status), " (", status, ")");
}

// Every op opens an OpScope on one of its tensors: the HIP device becomes that tensor's device for the duration of the
// op (kernels, torch's current stream and any at::empty are then on the GPU that owns the data, whatever the caller's
// current device is — model.to("cuda:1") with device 0 current), and every tensor checked afterwards must live there too.
thread_local const c10::Device* tl_op_device = nullptr;
struct OpScope {
  c10::Device dev;
  c10::DeviceGuard guard;   // generic guard: resolves to the (CUDA-masquerading) HIP implementation registered for the tensor's device type
  explicit OpScope(const Tensor& t) : dev(t.device()), guard(t.device()) {
    TORCH_CHECK(t.is_cuda(), "pfk ops need GPU tensors");
    tl_op_device = &dev;
  }
  ~OpScope() { tl_op_device = nullptr; }
};

pfk_stream_t cur_stream() { return static_cast<pfk_stream_t>(c10::hip::getCurrentHIPStream().stream()); }

void check_dev(const Tensor& t, const char* name) {
  TORCH_CHECK(t.is_cuda(), name, " must be a GPU tensor");
  TORCH_CHECK(tl_op_device == nullptr || t.device() == *tl_op_device, name, " is on ", t.device(), " but the op runs on ", *tl_op_device);
}

void check_dev_f32(const Tensor& t, const char* name) {
  check_dev(t, name);
  TORCH_CHECK(t.scalar_type() == at::kFloat, name, " must be float32");
}

// 2-D row-strided view [M, C] with unit channel stride.
void check_pm(const Tensor& t, const char* name) {
  check_dev_f32(t, name);
  TORCH_CHECK(t.dim() == 2 && (t.size(1) == 1 || t.stride(1) == 1), name,
              " must be a [pixels, channels] view with contiguous channels");
}

float* fptr(const Tensor& t) { return t.data_ptr<float>(); }

// [pixels, channels] view of fp32 or bf16 elements; returns true for bf16
bool check_pm_any(const Tensor& t, const char* name) {
  check_dev(t, name);
  TORCH_CHECK(t.scalar_type() == at::kFloat || t.scalar_type() == at::kBFloat16, name, " must be float32 or bfloat16");
  TORCH_CHECK(t.dim() == 2 && (t.size(1) == 1 || t.stride(1) == 1), name, " must be a [pixels, channels] view with contiguous channels");
  return t.scalar_type() == at::kBFloat16;
}

void check_pm_b16(const Tensor& t, const char* name) {
  check_dev(t, name);
  TORCH_CHECK(t.scalar_type() == at::kBFloat16, name, " must be bfloat16");
  TORCH_CHECK(t.dim() == 2 && (t.size(1) == 1 || t.stride(1) == 1), name, " must be a [pixels, channels] view with contiguous channels");
}


// out[b,i,j] = scale * <f1[b,i,:], f2[b,j,:]>
void corr_volume(const Tensor& f1, const Tensor& f2, double scale, Tensor out) {
  OpScope scope(f1);
  check_dev_f32(f1, "f1"); check_dev_f32(f2, "f2"); check_dev_f32(out, "out");
  TORCH_CHECK(f1.dim() == 3 && f2.dim() == 3 && out.dim() == 3, "corr_volume: [B,N,D] inputs, [B,N1,N2] out");
  TORCH_CHECK(f1.is_contiguous() && f2.is_contiguous() && out.is_contiguous(), "corr_volume: contiguous tensors");
  const int B = f1.size(0), N1 = f1.size(1), D = f1.size(2), N2 = f2.size(1);
  TORCH_CHECK(f2.size(0) == B && f2.size(2) == D && out.size(0) == B && out.size(1) == N1 && out.size(2) == N2,
              "corr_volume: shape mismatch");
  check_ok(pfk_corr_volume_f32(fptr(f1), D, fptr(f2), D, fptr(out), B, N1, N2, D, (float)scale, cur_stream()),
           "corr_volume");
}

void corr_pool2x2(const Tensor& in, Tensor out) {
  OpScope scope(in);
  check_dev(in, "in"); check_dev(out, "out");
  TORCH_CHECK(in.scalar_type() == out.scalar_type() && (in.scalar_type() == at::kFloat || in.scalar_type() == at::kBFloat16),
              "corr_pool2x2: float32 or bfloat16 maps");
  TORCH_CHECK(in.dim() == 3 && in.is_contiguous() && out.is_contiguous(), "corr_pool2x2: [M,H,W] contiguous");
  const int64_t M = in.size(0); const int H = in.size(1), W = in.size(2);
  TORCH_CHECK(out.numel() == M * (H / 2) * (W / 2), "corr_pool2x2: out has wrong size");
  if (out.numel() == 0) return;
  if (in.scalar_type() == at::kFloat) check_ok(pfk_corr_pool2x2_f32(fptr(in), fptr(out), M, H, W, cur_stream()), "corr_pool2x2");
  else check_ok(pfk_corr_pool2x2_bf16(in.data_ptr(), out.data_ptr(), M, H, W, cur_stream()), "corr_pool2x2 (bf16)");
}

// bf16 volume: f1, f2 bf16 [B,N,D] -> out bf16 [B,N1,N2]
void corr_volume_bf16(const Tensor& f1, const Tensor& f2, double scale, Tensor out) {
  OpScope scope(f1);
  check_dev(f1, "f1"); check_dev(f2, "f2"); check_dev(out, "out");
  TORCH_CHECK(f1.scalar_type() == at::kBFloat16 && f2.scalar_type() == at::kBFloat16 && out.scalar_type() == at::kBFloat16,
              "corr_volume_bf16: f1, f2 and out must be bfloat16");
  TORCH_CHECK(f1.dim() == 3 && f2.dim() == 3 && out.dim() == 3 && f1.is_contiguous() && f2.is_contiguous() && out.is_contiguous(),
              "corr_volume_bf16: contiguous [B,N,D] inputs, [B,N1,N2] out");
  const int B = f1.size(0), N1 = f1.size(1), D = f1.size(2), N2 = f2.size(1);
  TORCH_CHECK(f2.size(0) == B && f2.size(2) == D && out.size(0) == B && out.size(1) == N1 && out.size(2) == N2, "corr_volume_bf16: shape mismatch");
  check_ok(pfk_corr_volume_bf16(f1.data_ptr(), D, f2.data_ptr(), D, out.data_ptr(), B, N1, N2, D, (float)scale, cur_stream()), "corr_volume_bf16");
}

// pixel-major feature map [B*H*W, C] -> [B*(H/2)*(W/2), C]: 2x2 average == bilinear x0.5 (align_corners=False)
void fmap_pool2x2(const Tensor& in, Tensor out, int64_t B, int64_t H, int64_t W) {
  OpScope scope(in);
  check_pm(in, "in"); check_pm(out, "out");
  TORCH_CHECK(in.size(0) == B * H * W && out.size(0) == B * (H / 2) * (W / 2) && out.size(1) == in.size(1), "fmap_pool2x2: shapes");
  if (out.numel() == 0) return;
  check_ok(pfk_fmap_pool2x2_f32(fptr(in), in.stride(0), fptr(out), out.stride(0), (int)B, (int)H, (int)W, (int)in.size(1), cur_stream()),
           "fmap_pool2x2");
}

void corr_lookup(at::TensorList levels, const Tensor& coords, int64_t radius, Tensor out) {
  OpScope scope(coords);
  check_dev_f32(coords, "coords");
  const bool out_b16 = check_pm_any(out, "out");
  TORCH_CHECK(coords.dim() == 4 && coords.size(1) == 2 && coords.is_contiguous(), "corr_lookup: coords [B,2,h,w] contiguous");
  TORCH_CHECK(levels.size() >= 1 && levels.size() <= PFK_MAX_LEVELS, "corr_lookup: 1..8 levels");
  pfk_lookup_desc d{};
  d.B = coords.size(0); d.h = coords.size(2); d.w = coords.size(3);
  const int64_t M = (int64_t)d.B * d.h * d.w;
  const bool bf = levels[0].scalar_type() == at::kBFloat16;
  for (size_t l = 0; l < levels.size(); ++l) {
    const Tensor& v = levels[l];
    check_dev(v, "level");
    TORCH_CHECK(v.scalar_type() == (bf ? at::kBFloat16 : at::kFloat), "corr_lookup: levels must all be float32 or all bfloat16");
    TORCH_CHECK(v.dim() == 3 && v.is_contiguous() && v.size(0) == M, "corr_lookup: level must be [B*N,h_l,w_l] contiguous");
    d.levels[l] = v.data_ptr(); d.lvl_h[l] = v.size(1); d.lvl_w[l] = v.size(2);
  }
  d.num_levels = levels.size(); d.radius = radius;
  d.coords = fptr(coords); d.out = out.data_ptr(); d.out_ld = out.stride(0); d.out_bf16 = out_b16;
  TORCH_CHECK(out.size(0) == M, "corr_lookup: out rows");
  const int n = 2 * radius + 1;
  TORCH_CHECK(out.size(1) >= d.num_levels * n * n, "corr_lookup: out channels");
  check_ok(bf ? pfk_corr_lookup_bf16(&d, cur_stream()) : pfk_corr_lookup_f32(&d, cur_stream()), "corr_lookup");
}

void conv2d(at::TensorList srcs, int64_t B, int64_t H, int64_t W, int64_t kh, int64_t kw,
            const Tensor& weight, const c10::optional<Tensor>& bias, int64_t cout, int64_t epilogue,
            bool relu, double scale, const c10::optional<Tensor>& out, const c10::optional<Tensor>& h,
            const c10::optional<Tensor>& aux_z, const c10::optional<Tensor>& aux_rh,
            const c10::optional<Tensor>& workspace, const c10::optional<Tensor>& residual, int64_t stride,
            bool relu_after_residual, int64_t cout_active, int64_t cout_split) {
  OpScope scope(weight);
  TORCH_CHECK(srcs.size() >= 1 && srcs.size() <= 3, "conv2d: 1..3 sources");
  TORCH_CHECK(stride >= 1, "conv2d: stride");
  pfk_conv_desc d{};
  d.stride = (int)stride;
  d.cout_active = (int)cout_active;
  d.cout_split = (int)cout_split;
  d.relu_after_residual = relu_after_residual;
  const int64_t M = B * ((H - 1) / stride + 1) * ((W - 1) / stride + 1);   // output rows
  for (size_t i = 0; i < srcs.size(); ++i) {
    check_pm(srcs[i], "src");
    TORCH_CHECK(srcs[i].size(0) == B * H * W, "conv2d: src rows != B*H*W");
    d.src[i].ptr = fptr(srcs[i]); d.src[i].ld = srcs[i].stride(0); d.src[i].channels = srcs[i].size(1);
  }
  d.num_src = srcs.size();
  d.B = B; d.H = H; d.W = W; d.kh = kh; d.kw = kw; d.cout = cout;
  d.epilogue = epilogue; d.relu = relu; d.scale = (float)scale;
  // fp32 packed weight [cout, ktot] -> fp32 MFMA path; bf16 planes [nsplit, cout, ktot64] -> split-bf16 path
  const bool split = weight.scalar_type() == at::kBFloat16;
  if (split) {
    check_dev(weight, "weight");
    TORCH_CHECK(weight.is_contiguous() && weight.dim() == 3 && weight.size(1) == cout &&
                weight.size(0) >= 1 && weight.size(0) <= 3, "conv2d: bf16 weight planes [nsplit, cout, ktot]");
    TORCH_CHECK(weight.size(2) == pfk_conv_ktot_bf16(&d), "conv2d: bf16 weight planes have ktot ", weight.size(2),
                ", expected ", pfk_conv_ktot_bf16(&d));
  } else {
    check_dev_f32(weight, "weight");
    TORCH_CHECK(weight.is_contiguous() && weight.dim() == 2 && weight.size(0) == cout, "conv2d: packed weight [cout, ktot]");
    TORCH_CHECK(weight.size(1) == pfk_conv_ktot(&d), "conv2d: packed weight has ktot ", weight.size(1), ", expected ", pfk_conv_ktot(&d));
    d.weight = fptr(weight);
  }
  if (bias.has_value()) { check_dev_f32(*bias, "bias"); TORCH_CHECK(bias->numel() == cout && bias->is_contiguous()); d.bias = fptr(*bias); }
  if (out.has_value()) {
    check_pm(*out, "out");
    TORCH_CHECK(out->size(0) == M && out->size(1) == cout, "conv2d: out must be a [M, cout] view");
    d.out = fptr(*out); d.out_ld = out->stride(0); d.out_coff = 0;
  }
  if (h.has_value()) { check_pm(*h, "h"); TORCH_CHECK(h->size(0) == M); d.h = fptr(*h); d.h_ld = h->stride(0); }
  if (aux_z.has_value()) { check_dev_f32(*aux_z, "aux_z"); TORCH_CHECK(aux_z->is_contiguous()); d.aux_z = fptr(*aux_z); }
  if (aux_rh.has_value()) { check_dev_f32(*aux_rh, "aux_rh"); TORCH_CHECK(aux_rh->is_contiguous()); d.aux_rh = fptr(*aux_rh); }
  if (residual.has_value()) {
    check_pm(*residual, "residual");
    TORCH_CHECK(residual->size(0) == M && residual->size(1) == cout, "conv2d: residual must be a [M, cout] view");
    d.residual = fptr(*residual); d.residual_ld = residual->stride(0);
  }
  if (workspace.has_value()) {
    check_dev(*workspace, "workspace");
    TORCH_CHECK(workspace->is_contiguous(), "conv2d: workspace must be a contiguous GPU tensor");
    d.workspace = workspace->data_ptr(); d.workspace_bytes = (long long)workspace->nbytes();
  }
  if (split) check_ok(pfk_conv2d_bf16s(&d, weight.data_ptr(), (int)weight.size(0), cur_stream()), "conv2d (split bf16)");
  else check_ok(pfk_conv2d_f32(&d, cur_stream()), "conv2d");
}

// Grouped launch (pfk_conv2d_group_f32): problem i = one fp32 source srcs[i] ([B*H*W, cin_i] view), k[i] x k[i] taps, stride 1, packed
// weight weights[i] [cout_i, ktot_i], bias biases[i] (an empty tensor = none), relu[i], scale[i], out = outs[i] ([B*H*W, cout_i] view),
// residuals[i] (optional; an empty tensor = none): added after relu / scale
void conv2d_group(at::TensorList srcs, int64_t B, int64_t H, int64_t W, at::IntArrayRef k, at::TensorList weights, at::TensorList biases,
                  at::IntArrayRef relu, at::ArrayRef<double> scale, at::TensorList outs, at::TensorList residuals) {
  const size_t n = srcs.size();
  TORCH_CHECK(n >= 1 && n <= PFK_CONV_GROUP_MAX, "conv2d_group: 1..", PFK_CONV_GROUP_MAX, " problems");
  TORCH_CHECK(k.size() == n && weights.size() == n && biases.size() == n && relu.size() == n && scale.size() == n && outs.size() == n &&
              (residuals.size() == n || residuals.size() == 0), "conv2d_group: one entry per problem in every list (residuals: n or none)");
  OpScope scope(srcs[0]);
  pfk_conv_desc d[PFK_CONV_GROUP_MAX] = {};
  for (size_t i = 0; i < n; ++i) {
    check_pm(srcs[i], "src"); check_pm(outs[i], "out"); check_dev_f32(weights[i], "weight");
    TORCH_CHECK(srcs[i].size(0) == B * H * W && outs[i].size(0) == B * H * W, "conv2d_group: rows != B*H*W");
    const int cout = outs[i].size(1);
    d[i].num_src = 1;
    d[i].src[0].ptr = fptr(srcs[i]); d[i].src[0].ld = srcs[i].stride(0); d[i].src[0].channels = srcs[i].size(1);
    d[i].B = B; d[i].H = H; d[i].W = W; d[i].kh = k[i]; d[i].kw = k[i]; d[i].cout = cout; d[i].stride = 1;
    d[i].epilogue = PFK_EPI_LINEAR; d[i].relu = relu[i] != 0; d[i].scale = (float)scale[i];
    TORCH_CHECK(weights[i].is_contiguous() && weights[i].dim() == 2 && weights[i].size(0) == cout && weights[i].size(1) == pfk_conv_ktot(&d[i]),
                "conv2d_group: packed weight [cout, ktot] of problem ", i);
    d[i].weight = fptr(weights[i]);
    if (biases[i].numel() > 0) {
      check_dev_f32(biases[i], "bias");
      TORCH_CHECK(biases[i].numel() == cout && biases[i].is_contiguous(), "conv2d_group: bias [cout]");
      d[i].bias = fptr(biases[i]);
    }
    d[i].out = fptr(outs[i]); d[i].out_ld = outs[i].stride(0); d[i].out_coff = 0;
    if (residuals.size() == n && residuals[i].numel() > 0) {      // out = residual + scale * relu?(conv + bias)  (an empty tensor = none)
      check_pm(residuals[i], "residual");
      TORCH_CHECK(residuals[i].size(0) == B * H * W && residuals[i].size(1) == cout, "conv2d_group: residual must be a [M, cout] view");
      d[i].residual = fptr(residuals[i]); d[i].residual_ld = residuals[i].stride(0);
    }
  }
  check_ok(pfk_conv2d_group_f32(d, (int)n, cur_stream()), "conv2d_group");
}

// K8b: bf16 sources / weight (pfk_conv2d_b16); `out` may be bf16 or fp32, h / aux_z / residual fp32, aux_rh / h_b16 bf16
void conv2d_b16(at::TensorList srcs, int64_t B, int64_t H, int64_t W, int64_t kh, int64_t kw, const Tensor& weight,
                const c10::optional<Tensor>& bias, int64_t cout, int64_t epilogue, bool relu, double scale,
                const c10::optional<Tensor>& out, const c10::optional<Tensor>& h, const c10::optional<Tensor>& h_b16,
                const c10::optional<Tensor>& aux_z, const c10::optional<Tensor>& aux_rh, const c10::optional<Tensor>& residual,
                int64_t stride, bool relu_after_residual) {
  OpScope scope(weight);
  if (weight.dim() == 3) {
    // batched GEMM (pfk_conv_b16_desc.batches): src [Bt, M, K] bf16, weight [Bt, cout, K] bf16, out [Bt, M, cout], residual [Bt, M, cout]
    TORCH_CHECK(srcs.size() == 1 && srcs[0].dim() == 3 && kh == 1 && kw == 1 && epilogue == PFK_EPI_LINEAR && out.has_value() && out->dim() == 3,
                "conv2d_b16 (batched): one [Bt, M, K] source, [Bt, cout, K] weight, [Bt, M, cout] out, 1x1, linear epilogue");
    const Tensor& a = srcs[0];
    check_dev(a, "src"); check_dev(weight, "weight"); check_dev(*out, "out");
    const int64_t Bt = a.size(0), M = a.size(1), K = a.size(2);
    TORCH_CHECK(a.scalar_type() == at::kBFloat16 && weight.scalar_type() == at::kBFloat16 && a.stride(2) == 1 && weight.is_contiguous() &&
                weight.size(0) == Bt && weight.size(1) == cout && weight.size(2) == (K + 63) / 64 * 64 && B == 1 && H * W == M &&
                out->size(0) == Bt && out->size(1) == M && out->size(2) == cout && out->stride(2) == 1, "conv2d_b16 (batched): shapes");
    pfk_conv_b16_desc d{};
    d.src[0].ptr = a.data_ptr(); d.src[0].ld = a.stride(1); d.src[0].channels = K; d.num_src = 1;
    d.B = 1; d.H = H; d.W = W; d.kh = 1; d.kw = 1; d.cout = cout; d.epilogue = PFK_EPI_LINEAR; d.relu = relu; d.scale = (float)scale;
    d.weight = weight.data_ptr();
    d.out = out->data_ptr(); d.out_ld = out->stride(1); d.out_bf16 = out->scalar_type() == at::kBFloat16;
    TORCH_CHECK(d.out_bf16 || out->scalar_type() == at::kFloat, "conv2d_b16: out must be bfloat16 or float32");
    d.batches = Bt; d.src_batch_stride = a.stride(0); d.weight_batch_stride = weight.stride(0); d.out_batch_stride = out->stride(0);
    if (bias.has_value()) { check_dev_f32(*bias, "bias"); TORCH_CHECK(bias->numel() == cout && bias->is_contiguous()); d.bias = fptr(*bias); }
    if (residual.has_value()) {
      check_dev(*residual, "residual");
      TORCH_CHECK(residual->dim() == 3 && residual->size(0) == Bt && residual->size(1) == M && residual->size(2) == cout && residual->stride(2) == 1,
                  "conv2d_b16 (batched): residual [Bt, M, cout]");
      d.residual_bf16 = residual->scalar_type() == at::kBFloat16;
      TORCH_CHECK(d.residual_bf16 || residual->scalar_type() == at::kFloat, "conv2d_b16: residual must be bfloat16 or float32");
      d.residual = residual->data_ptr(); d.residual_ld = residual->stride(1); d.residual_batch_stride = residual->stride(0);
    }
    check_ok(pfk_conv2d_b16(&d, cur_stream()), "conv2d_b16 (batched)");
    return;
  }
  TORCH_CHECK(srcs.size() >= 1 && srcs.size() <= 3, "conv2d_b16: 1..3 sources");
  TORCH_CHECK(stride >= 1, "conv2d_b16: stride");
  pfk_conv_b16_desc d{};
  d.stride = (int)stride;
  d.relu_after_residual = relu_after_residual;
  const int64_t M = B * ((H - 1) / stride + 1) * ((W - 1) / stride + 1);
  for (size_t i = 0; i < srcs.size(); ++i) {
    check_pm_b16(srcs[i], "src");
    TORCH_CHECK(srcs[i].size(0) == B * H * W, "conv2d_b16: src rows != B*H*W");
    d.src[i].ptr = srcs[i].data_ptr(); d.src[i].ld = srcs[i].stride(0); d.src[i].channels = srcs[i].size(1);
  }
  d.num_src = srcs.size();
  d.B = B; d.H = H; d.W = W; d.kh = kh; d.kw = kw; d.cout = cout;
  d.epilogue = epilogue; d.relu = relu; d.scale = (float)scale;
  check_dev(weight, "weight");
  TORCH_CHECK(weight.scalar_type() == at::kBFloat16 && weight.is_contiguous() && weight.dim() == 2 && weight.size(0) == cout,
              "conv2d_b16: bf16 packed weight [cout, ktot]");
  TORCH_CHECK(weight.size(1) == pfk_conv_ktot_b16(&d), "conv2d_b16: packed weight has ktot ", weight.size(1), ", expected ",
              pfk_conv_ktot_b16(&d));
  d.weight = weight.data_ptr();
  if (bias.has_value()) { check_dev_f32(*bias, "bias"); TORCH_CHECK(bias->numel() == cout && bias->is_contiguous()); d.bias = fptr(*bias); }
  if (out.has_value()) {
    check_dev(*out, "out");
    TORCH_CHECK(out->dim() == 2 && (out->size(1) == 1 || out->stride(1) == 1) && out->size(0) == M && out->size(1) == cout,
                "conv2d_b16: out must be a [M, cout] view");
    TORCH_CHECK(out->scalar_type() == at::kBFloat16 || out->scalar_type() == at::kFloat, "conv2d_b16: out must be bfloat16 or float32");
    d.out = out->data_ptr(); d.out_ld = out->stride(0); d.out_coff = 0; d.out_bf16 = out->scalar_type() == at::kBFloat16;
  }
  if (h.has_value()) { check_pm(*h, "h"); TORCH_CHECK(h->size(0) == M); d.h = fptr(*h); d.h_ld = h->stride(0); }
  if (h_b16.has_value()) { check_pm_b16(*h_b16, "h_b16"); TORCH_CHECK(h_b16->size(0) == M); d.h_b16 = h_b16->data_ptr(); d.h_b16_ld = h_b16->stride(0); }
  if (aux_z.has_value()) {
    check_dev(*aux_z, "aux_z");
    TORCH_CHECK(aux_z->scalar_type() == at::kBFloat16 && aux_z->is_contiguous(), "conv2d_b16: aux_z must be contiguous bfloat16");
    d.aux_z = aux_z->data_ptr();
  }
  if (aux_rh.has_value()) {
    check_dev(*aux_rh, "aux_rh");
    TORCH_CHECK(aux_rh->scalar_type() == at::kBFloat16 && aux_rh->is_contiguous(), "conv2d_b16: aux_rh must be contiguous bfloat16");
    d.aux_rh = aux_rh->data_ptr();
  }
  if (residual.has_value()) {
    const bool r16 = check_pm_any(*residual, "residual");
    TORCH_CHECK(epilogue == PFK_EPI_LINEAR || r16, "conv2d_b16: the GRU epilogues take a bfloat16 residual (the context term)");
    TORCH_CHECK(residual->size(0) == M && residual->size(1) == cout, "conv2d_b16: residual must be a [M, cout] view");
    d.residual = residual->data_ptr(); d.residual_ld = residual->stride(0); d.residual_bf16 = r16 && epilogue == PFK_EPI_LINEAR;
  }
  check_ok(pfk_conv2d_b16(&d, cur_stream()), "conv2d_b16");
}

void debug_set_b16(int64_t cfg) { check_ok(pfk_debug_set_b16((int)cfg), "debug_set_b16"); }
void debug_set_stem_valu(int64_t on) { check_ok(pfk_debug_set_stem_valu((int)on), "debug_set_stem_valu"); }
void debug_set_cin2_valu(int64_t on) { check_ok(pfk_debug_set_cin2_valu((int)on), "debug_set_cin2_valu"); }

void conv_cin2(const Tensor& in, const Tensor& weight, const c10::optional<Tensor>& bias, Tensor out,
               int64_t B, int64_t H, int64_t W, int64_t k, bool relu) {
  OpScope scope(in);
  check_pm(in, "in"); check_dev_f32(weight, "weight");
  const bool out_b16 = check_pm_any(out, "out");
  const int cout = out.size(1);
  TORCH_CHECK(weight.is_contiguous() && weight.numel() == k * k * 2 * cout, "conv_cin2: weight [k*k,2,cout]");
  TORCH_CHECK(in.size(0) == B * H * W && out.size(0) == B * H * W && in.size(1) >= 2);
  const float* bp = nullptr;
  if (bias.has_value()) { check_dev_f32(*bias, "bias"); bp = fptr(*bias); }
  if (out_b16) check_ok(pfk_conv_cin2_b16(fptr(in), in.stride(0), fptr(weight), bp, out.data_ptr(), out.stride(0), 0, B, H, W, k,
                                          cout, relu, cur_stream()), "conv_cin2 (bf16 out)");
  else check_ok(pfk_conv_cin2_f32(fptr(in), in.stride(0), fptr(weight), bp, fptr(out), out.stride(0), 0, B, H, W, k,
                                  cout, relu, cur_stream()), "conv_cin2");
}

void flow_delta(const Tensor& in, const Tensor& weight, const c10::optional<Tensor>& bias, const Tensor& coords0,
                Tensor coords1, const c10::optional<Tensor>& delta_out, const c10::optional<Tensor>& flow_out,
                const c10::optional<Tensor>& flow_out_b16) {
  OpScope scope(in);
  const bool in_b16 = check_pm_any(in, "in");
  check_dev_f32(weight, "weight"); check_dev_f32(coords0, "coords0"); check_dev_f32(coords1, "coords1");
  TORCH_CHECK(coords0.dim() == 4 && coords0.size(1) == 2 && coords0.is_contiguous() && coords1.is_contiguous() &&
              coords1.sizes() == coords0.sizes(), "flow_delta: coords [B,2,h,w] contiguous");
  const int B = coords0.size(0), H = coords0.size(2), W = coords0.size(3), cin = in.size(1);
  TORCH_CHECK(in.size(0) == (int64_t)B * H * W && weight.is_contiguous() && weight.numel() == 9 * 2 * cin);
  const float* bp = nullptr; float* dp = nullptr; float* fp = nullptr; int fld = 0;
  if (bias.has_value()) { check_dev_f32(*bias, "bias"); bp = fptr(*bias); }
  if (delta_out.has_value()) { check_dev_f32(*delta_out, "delta"); TORCH_CHECK(delta_out->is_contiguous() && delta_out->sizes() == coords0.sizes()); dp = fptr(*delta_out); }
  if (flow_out.has_value()) { check_pm(*flow_out, "flow_out"); TORCH_CHECK(flow_out->size(1) >= 2); fp = fptr(*flow_out); fld = flow_out->stride(0); }
  if (in_b16) {
    void* fb = nullptr; int fbld = 0;
    if (flow_out_b16.has_value()) { check_pm_b16(*flow_out_b16, "flow_out_b16"); TORCH_CHECK(flow_out_b16->size(1) >= 2); fb = flow_out_b16->data_ptr(); fbld = flow_out_b16->stride(0); }
    check_ok(pfk_flow_delta_b16(in.data_ptr(), in.stride(0), cin, fptr(weight), bp, fptr(coords0), fptr(coords1), dp, fp, fld, fb, fbld,
                                B, H, W, cur_stream()), "flow_delta (bf16 in)");
    return;
  }
  TORCH_CHECK(!flow_out_b16.has_value(), "flow_delta: flow_out_b16 goes with a bfloat16 input");
  check_ok(pfk_flow_delta_f32(fptr(in), in.stride(0), cin, fptr(weight), bp, fptr(coords0), fptr(coords1), dp, fp,
                              fld, B, H, W, cur_stream()), "flow_delta");
}

void flow_from_coords(const Tensor& coords0, const Tensor& coords1, Tensor flow_out) {
  OpScope scope(coords0);
  check_dev_f32(coords0, "coords0"); check_dev_f32(coords1, "coords1"); check_pm(flow_out, "flow_out");
  TORCH_CHECK(coords0.dim() == 4 && coords0.is_contiguous() && coords1.is_contiguous() && coords1.sizes() == coords0.sizes());
  check_ok(pfk_flow_from_coords_f32(fptr(coords0), fptr(coords1), fptr(flow_out), flow_out.stride(0), coords0.size(0),
                                    coords0.size(2), coords0.size(3), cur_stream()), "flow_from_coords");
}

void convex_upsample(const Tensor& flow, const Tensor& mask, Tensor out) {
  OpScope scope(flow);
  check_dev_f32(flow, "flow"); check_pm(mask, "mask"); check_dev_f32(out, "out");
  TORCH_CHECK(flow.dim() == 4 && flow.size(1) == 2 && flow.is_contiguous() && out.is_contiguous());
  const int B = flow.size(0), H = flow.size(2), W = flow.size(3);
  TORCH_CHECK(mask.size(0) == (int64_t)B * H * W && mask.size(1) == 576 && out.numel() == (int64_t)B * 2 * 64 * H * W);
  check_ok(pfk_convex_upsample_f32(fptr(flow), fptr(mask), mask.stride(0), fptr(out), B, H, W, cur_stream()),
           "convex_upsample");
}

void upflow8(const Tensor& coords0, const Tensor& coords1, Tensor out) {
  OpScope scope(coords0);
  check_dev_f32(coords0, "coords0"); check_dev_f32(coords1, "coords1"); check_dev_f32(out, "out");
  TORCH_CHECK(coords0.dim() == 4 && coords0.size(1) == 2 && coords0.is_contiguous() && coords1.is_contiguous() &&
              coords1.sizes() == coords0.sizes(), "upflow8: coords [B, 2, H, W] contiguous");
  const int64_t B = coords0.size(0), H = coords0.size(2), W = coords0.size(3);
  TORCH_CHECK(out.is_contiguous() && out.dim() == 4 && out.size(0) == B && out.size(1) == 2 && out.size(2) == 8 * H && out.size(3) == 8 * W,
              "upflow8: out [B, 2, 8H, 8W] contiguous");
  check_ok(pfk_upflow8_f32(fptr(coords0), fptr(coords1), fptr(out), (int)B, (int)H, (int)W, cur_stream()), "upflow8");
}

void convex_upsample_pm(const Tensor& flow_pm, const Tensor& mask, Tensor out) {
  OpScope scope(flow_pm);
  check_pm(flow_pm, "flow_pm"); check_dev_f32(out, "out");
  const bool mask_b16 = check_pm_any(mask, "mask");
  TORCH_CHECK(out.dim() == 4 && out.size(1) == 2 && out.is_contiguous() && out.size(2) % 8 == 0 && out.size(3) % 8 == 0);
  const int B = out.size(0), H = out.size(2) / 8, W = out.size(3) / 8;
  TORCH_CHECK(mask.size(0) == (int64_t)B * H * W && mask.size(1) == 576 && flow_pm.size(0) == mask.size(0) && flow_pm.size(1) >= 2);
  if (mask_b16) {
    check_ok(pfk_convex_upsample_pm_b16(fptr(flow_pm), flow_pm.stride(0), mask.data_ptr(), mask.stride(0), fptr(out), B, H, W, cur_stream()),
             "convex_upsample_pm (bf16 mask)");
    return;
  }
  check_ok(pfk_convex_upsample_pm_f32(fptr(flow_pm), flow_pm.stride(0), fptr(mask), mask.stride(0), fptr(out), B, H, W,
                                      cur_stream()), "convex_upsample_pm");
}

// fused mask conv2 + softmax + convex upsampling: x [M, cin] (a view of the fh|mask hidden buffer), weight / bias row-permuted (pfk.h)
void mask_upsample(const Tensor& x, const Tensor& weight_perm, const c10::optional<Tensor>& bias_perm, double scale,
                   const Tensor& flow_pm, Tensor out) {
  OpScope scope(x);
  const bool b16 = check_pm_any(x, "x");
  check_pm(flow_pm, "flow_pm"); check_dev_f32(out, "out");
  if (b16) check_dev(weight_perm, "weight"); else check_dev_f32(weight_perm, "weight");
  TORCH_CHECK(!b16 || weight_perm.scalar_type() == at::kBFloat16, "mask_upsample: a bf16 activation needs the bf16 permuted weight");
  TORCH_CHECK(out.dim() == 4 && out.size(1) == 2 && out.is_contiguous() && out.size(2) % 8 == 0 && out.size(3) % 8 == 0, "mask_upsample: out [B,2,8H,8W]");
  const int B = out.size(0), H = out.size(2) / 8, W = out.size(3) / 8;
  const int cin = x.size(1);
  TORCH_CHECK(x.size(0) == (int64_t)B * H * W && flow_pm.size(0) == x.size(0) && flow_pm.size(1) >= 2, "mask_upsample: rows");
  TORCH_CHECK(weight_perm.is_contiguous() && weight_perm.dim() == 2 && weight_perm.size(0) == 640 && weight_perm.size(1) == cin, "mask_upsample: weight [640, cin] (packing.permute_mask_head)");
  const float* bp = nullptr;
  if (bias_perm.has_value()) { check_dev_f32(*bias_perm, "bias"); TORCH_CHECK(bias_perm->numel() == 640 && bias_perm->is_contiguous()); bp = fptr(*bias_perm); }
  if (b16) {
    check_ok(pfk_mask_upsample_b16(x.data_ptr(), x.stride(0), cin, weight_perm.data_ptr(), bp, (float)scale, fptr(flow_pm), flow_pm.stride(0),
                                   fptr(out), B, H, W, cur_stream()), "mask_upsample (bf16)");
    return;
  }
  check_ok(pfk_mask_upsample_f32(fptr(x), x.stride(0), cin, fptr(weight_perm), bp, (float)scale, fptr(flow_pm), flow_pm.stride(0),
                                 fptr(out), B, H, W, cur_stream()), "mask_upsample");
}

// alt_cuda_corr.forward semantics (correlation.cpp:23-37): returns [B, N, (2r+1)^2, H1, W1], unscaled
Tensor altcorr_forward(const Tensor& fmap1, const Tensor& fmap2, const Tensor& coords, int64_t radius) {
  OpScope scope(fmap1);
  // fp32 maps (the reference extension's only type, correlation_kernel.cu:275) or bf16 maps (pfk_altcorr_forward_bf16: exact widening,
  // fp32 products and accumulation; the result is fp32 either way — coords stay fp32)
  const bool bf = fmap1.scalar_type() == at::kBFloat16;
  if (bf) {
    TORCH_CHECK(fmap1.is_cuda() && fmap2.is_cuda() && fmap2.scalar_type() == at::kBFloat16, "altcorr_forward: both feature maps bf16 on the GPU");
  } else {
    check_dev_f32(fmap1, "fmap1"); check_dev_f32(fmap2, "fmap2");
  }
  check_dev_f32(coords, "coords");
  TORCH_CHECK(fmap1.dim() == 4 && fmap2.dim() == 4 && coords.dim() == 5 && coords.size(4) == 2, "altcorr_forward: fmap [B,H,W,C], coords [B,N,H1,W1,2]");
  TORCH_CHECK(fmap1.is_contiguous() && fmap2.is_contiguous() && coords.is_contiguous(), "altcorr_forward: contiguous inputs");  // CHECK_CONTIGUOUS in the reference
  const int B = fmap1.size(0), H1 = fmap1.size(1), W1 = fmap1.size(2), C = fmap1.size(3);
  const int H2 = fmap2.size(1), W2 = fmap2.size(2), N = coords.size(1);
  TORCH_CHECK(fmap2.size(0) == B && fmap2.size(3) == C && coords.size(0) == B && coords.size(2) == H1 && coords.size(3) == W1);
  const int rd = 2 * radius + 1;
  Tensor out = at::empty({B, N, rd * rd, H1, W1}, coords.options());
  // the launch-wide gate's per-block counts (pfk.h): a few hundred bytes from the caching allocator, written before they are read
  Tensor ws = at::empty({std::max<int64_t>(1, pfk_altcorr_workspace_bytes(B, H1, W1) / 4)}, coords.options().dtype(at::kInt));
  for (int n = 0; n < N; ++n) {
    Tensor cn = coords.select(1, n).contiguous();
    Tensor on = N == 1 ? out.view({B, rd * rd, H1, W1}) : at::empty({B, rd * rd, H1, W1}, coords.options());
    if (bf)
      check_ok(pfk_altcorr_forward_bf16(fmap1.data_ptr(), fmap2.data_ptr(), fptr(cn), fptr(on), B, H1, W1, H2, W2, C, radius, ws.data_ptr(), cur_stream()),
               "altcorr_forward (bf16 maps)");
    else
      check_ok(pfk_altcorr_forward_f32(fptr(fmap1), fptr(fmap2), fptr(cn), fptr(on), B, H1, W1, H2, W2, C, radius, ws.data_ptr(), cur_stream()),
               "altcorr_forward");
    if (N != 1) out.select(1, n).copy_(on);
  }
  return out;
}

// alt_cuda_corr.backward semantics (correlation.cpp:39-49): returns {fmap1_grad, fmap2_grad, coords_grad (zeros)}
std::vector<Tensor> altcorr_backward(const Tensor& fmap1, const Tensor& fmap2, const Tensor& coords, const Tensor& corr_grad,
                                     int64_t radius) {
  OpScope scope(fmap1);
  check_dev_f32(fmap1, "fmap1"); check_dev_f32(fmap2, "fmap2"); check_dev_f32(coords, "coords"); check_dev_f32(corr_grad, "corr_grad");
  TORCH_CHECK(fmap1.is_contiguous() && fmap2.is_contiguous() && coords.is_contiguous() && corr_grad.is_contiguous(),
              "altcorr_backward: contiguous inputs");
  TORCH_CHECK(fmap1.dim() == 4 && fmap2.dim() == 4 && coords.dim() == 5 && corr_grad.dim() == 5);
  const int B = fmap1.size(0), H1 = fmap1.size(1), W1 = fmap1.size(2), C = fmap1.size(3);
  const int H2 = fmap2.size(1), W2 = fmap2.size(2), N = coords.size(1);
  const int rd = 2 * radius + 1;
  TORCH_CHECK(corr_grad.size(0) == B && corr_grad.size(1) == N && corr_grad.size(2) == rd * rd && corr_grad.size(3) == H1 &&
              corr_grad.size(4) == W1, "altcorr_backward: corr_grad [B,N,(2r+1)^2,H1,W1]");
  Tensor g1 = at::empty_like(fmap1), g2 = at::empty_like(fmap2), gc = at::zeros_like(coords);
  // correlation_kernel.cu:122-256 loops over the N coordinate sets inside the kernel; here one launch per set, summed
  for (int n = 0; n < N; ++n) {
    Tensor cn = N == 1 ? coords.view({B, H1, W1, 2}) : coords.select(1, n).contiguous();
    Tensor gn = N == 1 ? corr_grad.view({B, rd * rd, H1, W1}) : corr_grad.select(1, n).contiguous();
    Tensor t1 = n == 0 ? g1 : at::empty_like(fmap1), t2 = n == 0 ? g2 : at::empty_like(fmap2);
    check_ok(pfk_altcorr_backward_f32(fptr(fmap1), fptr(fmap2), fptr(cn), fptr(gn), fptr(t1), fptr(t2), B, H1, W1, H2, W2, C, radius,
                                      cur_stream()), "altcorr_backward");
    if (n) { g1.add_(t1); g2.add_(t2); }
  }
  return {g1, g2, gc};
}

// ---- blocked volume layout (include/pfk.h "K1-K3 on the BLOCKED volume layout"): levels are [M, pfk_blocked_map_elems(h_l, w_l)]
void fmap_to_blocked(const Tensor& in, Tensor out, int64_t B, int64_t H, int64_t W) {
  OpScope scope(in);
  check_pm(in, "in"); check_pm(out, "out");
  TORCH_CHECK(in.size(0) == B * H * W && out.size(0) == B * pfk_blocked_map_elems((int)H, (int)W) && out.size(1) == in.size(1),
              "fmap_to_blocked: in [B*H*W, C], out [B*blocked_map_elems(H, W), C]");
  check_ok(pfk_fmap_to_blocked_f32(fptr(in), in.stride(0), fptr(out), out.stride(0), (int)B, (int)H, (int)W, (int)in.size(1), cur_stream()),
           "fmap_to_blocked");
}

void corr_pool2x2_blocked(const Tensor& in, Tensor out, int64_t H, int64_t W) {
  OpScope scope(in);
  check_dev(in, "in"); check_dev(out, "out");
  TORCH_CHECK(in.scalar_type() == out.scalar_type() && (in.scalar_type() == at::kFloat || in.scalar_type() == at::kBFloat16),
              "corr_pool2x2_blocked: float32 or bfloat16 maps");
  TORCH_CHECK(in.dim() == 2 && out.dim() == 2 && in.is_contiguous() && out.is_contiguous() && in.size(0) == out.size(0),
              "corr_pool2x2_blocked: [M, elems] contiguous");
  TORCH_CHECK(in.size(1) == pfk_blocked_map_elems((int)H, (int)W) && out.size(1) == pfk_blocked_map_elems((int)(H / 2), (int)(W / 2)),
              "corr_pool2x2_blocked: map sizes");
  if (out.numel() == 0) return;
  if (in.scalar_type() == at::kFloat) check_ok(pfk_corr_pool2x2_blocked_f32(fptr(in), fptr(out), in.size(0), (int)H, (int)W, cur_stream()), "corr_pool2x2_blocked");
  else check_ok(pfk_corr_pool2x2_blocked_bf16(in.data_ptr(), out.data_ptr(), in.size(0), (int)H, (int)W, cur_stream()), "corr_pool2x2_blocked (bf16)");
}

void corr_lookup_blocked(at::TensorList levels, at::IntArrayRef lvl_h, at::IntArrayRef lvl_w, const Tensor& coords, int64_t radius, Tensor out) {
  OpScope scope(coords);
  check_dev_f32(coords, "coords");
  const bool out_b16 = check_pm_any(out, "out");
  TORCH_CHECK(coords.dim() == 4 && coords.size(1) == 2 && coords.is_contiguous(), "corr_lookup_blocked: coords [B,2,h,w] contiguous");
  TORCH_CHECK(levels.size() >= 1 && levels.size() <= PFK_MAX_LEVELS && lvl_h.size() == levels.size() && lvl_w.size() == levels.size(),
              "corr_lookup_blocked: 1..8 levels with their map sizes");
  pfk_lookup_desc d{};
  d.B = coords.size(0); d.h = coords.size(2); d.w = coords.size(3);
  const int64_t M = (int64_t)d.B * d.h * d.w;
  const bool bf = levels[0].scalar_type() == at::kBFloat16;
  for (size_t l = 0; l < levels.size(); ++l) {
    const Tensor& v = levels[l];
    check_dev(v, "level");
    TORCH_CHECK(v.scalar_type() == (bf ? at::kBFloat16 : at::kFloat), "corr_lookup_blocked: levels must all be float32 or all bfloat16");
    TORCH_CHECK(lvl_h[l] > 0 && lvl_w[l] > 0 && v.dim() == 2 && v.is_contiguous() && v.size(0) == M &&
                v.size(1) == pfk_blocked_map_elems((int)lvl_h[l], (int)lvl_w[l]),
                "corr_lookup_blocked: level must be [B*N, blocked_map_elems(h_l, w_l)] contiguous");
    d.levels[l] = v.data_ptr(); d.lvl_h[l] = (int)lvl_h[l]; d.lvl_w[l] = (int)lvl_w[l];
  }
  d.num_levels = levels.size(); d.radius = radius;
  d.coords = fptr(coords); d.out = out.data_ptr(); d.out_ld = out.stride(0); d.out_bf16 = out_b16;
  TORCH_CHECK(out.size(0) == M, "corr_lookup_blocked: out rows");
  const int n = 2 * radius + 1;
  TORCH_CHECK(out.size(1) >= d.num_levels * n * n, "corr_lookup_blocked: out channels");
  check_ok(bf ? pfk_corr_lookup_blocked_bf16(&d, cur_stream()) : pfk_corr_lookup_blocked_f32(&d, cur_stream()), "corr_lookup_blocked");
}

int64_t blocked_map_elems(int64_t H, int64_t W) { return pfk_blocked_map_elems((int)H, (int)W); }

// backward of corr_lookup: accumulates into grad_levels[l] ([B*N, ld_l] buffers holding one [h_l][w_l] map per row)
void corr_lookup_bwd(at::TensorList grad_levels, at::IntArrayRef lvl_h, at::IntArrayRef lvl_w, const Tensor& coords,
                     int64_t radius, const Tensor& grad_out) {
  OpScope scope(coords);
  check_dev_f32(coords, "coords"); check_pm(grad_out, "grad_out");
  TORCH_CHECK(coords.dim() == 4 && coords.size(1) == 2 && coords.is_contiguous(), "corr_lookup_bwd: coords [B,2,h,w] contiguous");
  TORCH_CHECK(grad_levels.size() >= 1 && grad_levels.size() <= PFK_MAX_LEVELS && lvl_h.size() == grad_levels.size() &&
              lvl_w.size() == grad_levels.size(), "corr_lookup_bwd: 1..8 levels with their map sizes");
  pfk_lookup_bwd_desc d{};
  d.B = coords.size(0); d.h = coords.size(2); d.w = coords.size(3);
  const int64_t M = (int64_t)d.B * d.h * d.w;
  for (size_t l = 0; l < grad_levels.size(); ++l) {
    const Tensor& v = grad_levels[l];
    check_pm(v, "grad_level");
    TORCH_CHECK(v.size(0) == M && v.size(1) >= lvl_h[l] * lvl_w[l], "corr_lookup_bwd: grad level must be [B*N, >= h_l*w_l]");
    d.grad_levels[l] = fptr(v); d.lvl_h[l] = lvl_h[l]; d.lvl_w[l] = lvl_w[l]; d.lvl_ld[l] = v.stride(0);
  }
  d.num_levels = grad_levels.size(); d.radius = radius;
  d.coords = fptr(coords); d.grad_out = fptr(grad_out); d.grad_out_ld = grad_out.stride(0);
  const int n = 2 * radius + 1;
  TORCH_CHECK(grad_out.size(0) == M && grad_out.size(1) >= d.num_levels * n * n, "corr_lookup_bwd: grad_out shape");
  check_ok(pfk_corr_lookup_bwd_f32(&d, cur_stream()), "corr_lookup_bwd");
}

// backward of corr_volume for one pyramid level, all batch elements: dC [B,N1,ldc] (columns >= N2 zero), f1 [B,N1,D],
// f2cm [B,D,ld2cm] channel-major zero-padded -> df1 [B,N1,D] (+= when accumulate), df2 [B,ldc,Dpad]
void corr_volume_bwd(const Tensor& dC, int64_t N2, const Tensor& f1, const Tensor& f2cm, double scale, Tensor df1, bool accumulate,
                     Tensor df2) {
  OpScope scope(dC);
  check_dev_f32(dC, "dC"); check_dev_f32(f1, "f1"); check_dev_f32(f2cm, "f2cm"); check_dev_f32(df1, "df1"); check_dev_f32(df2, "df2");
  TORCH_CHECK(dC.dim() == 3 && f1.dim() == 3 && f2cm.dim() == 3 && df1.dim() == 3 && df2.dim() == 3, "corr_volume_bwd: 3-D tensors");
  TORCH_CHECK(dC.is_contiguous() && f1.is_contiguous() && f2cm.is_contiguous() && df1.is_contiguous() && df2.is_contiguous(),
              "corr_volume_bwd: contiguous tensors");
  const int B = dC.size(0), N1 = dC.size(1), ldc = dC.size(2), D = f1.size(2), ld2cm = f2cm.size(2);
  TORCH_CHECK(f1.size(0) == B && f1.size(1) == N1 && f2cm.size(0) == B && f2cm.size(1) == D && df1.sizes() == f1.sizes() &&
              df2.size(0) == B && df2.size(1) == ldc && df2.size(2) == (D + 31) / 32 * 32, "corr_volume_bwd: shape mismatch");
  const long long need = pfk_corr_volume_bwd_workspace_bytes(N1, ldc, D);
  Tensor ws;
  void* wsp = nullptr;
  if (need > 0) { ws = at::empty({(int64_t)need}, dC.options().dtype(at::kByte)); wsp = ws.data_ptr(); }
  for (int b = 0; b < B; ++b)
    check_ok(pfk_corr_volume_bwd_f32(fptr(dC) + (int64_t)b * N1 * ldc, ldc, N1, (int)N2, fptr(f1) + (int64_t)b * N1 * D, D,
                                     fptr(f2cm) + (int64_t)b * D * ld2cm, ld2cm, D, (float)scale, fptr(df1) + (int64_t)b * N1 * D, D,
                                     accumulate, fptr(df2) + (int64_t)b * ldc * df2.size(2), wsp, need, cur_stream()),
             "corr_volume_bwd");
}

// backward of convex_upsample(_pm): flow NCHW [B,2,H,W] or pixel-major [M, >=2]
void convex_upsample_bwd(const Tensor& flow, const Tensor& mask, const Tensor& grad_out, Tensor grad_mask, Tensor grad_flow) {
  OpScope scope(mask);
  check_dev_f32(flow, "flow"); check_pm(mask, "mask"); check_dev_f32(grad_out, "grad_out"); check_pm(grad_mask, "grad_mask");
  check_dev_f32(grad_flow, "grad_flow");
  TORCH_CHECK(grad_out.dim() == 4 && grad_out.size(1) == 2 && grad_out.is_contiguous() && grad_out.size(2) % 8 == 0 &&
              grad_out.size(3) % 8 == 0, "convex_upsample_bwd: grad_out [B,2,8H,8W] contiguous");
  const int B = grad_out.size(0), H = grad_out.size(2) / 8, W = grad_out.size(3) / 8;
  const int64_t M = (int64_t)B * H * W;
  int flow_ld = 0;
  if (flow.dim() == 2) { check_pm(flow, "flow"); TORCH_CHECK(flow.size(0) == M && flow.size(1) >= 2); flow_ld = flow.stride(0); }
  else TORCH_CHECK(flow.dim() == 4 && flow.is_contiguous() && flow.size(0) == B && flow.size(1) == 2 && flow.size(2) == H && flow.size(3) == W,
                   "convex_upsample_bwd: flow [B,2,H,W] contiguous or pixel-major [M,2+]");
  TORCH_CHECK(mask.size(0) == M && mask.size(1) == 576 && grad_mask.size(0) == M && grad_mask.size(1) == 576, "convex_upsample_bwd: mask shapes");
  TORCH_CHECK(grad_flow.is_contiguous() && grad_flow.numel() == M * 2, "convex_upsample_bwd: grad_flow [B,2,H,W]");
  const long long need = pfk_convex_upsample_bwd_workspace_bytes(B, H, W);
  Tensor ws = at::empty({(int64_t)need}, mask.options().dtype(at::kByte));
  check_ok(pfk_convex_upsample_bwd_f32(fptr(flow), flow_ld, fptr(mask), mask.stride(0), fptr(grad_out), fptr(grad_mask),
                                       grad_mask.stride(0), fptr(grad_flow), ws.data_ptr(), need, B, H, W, cur_stream()),
           "convex_upsample_bwd");
}

void nchw_to_pm(const Tensor& in, Tensor out) {
  OpScope scope(in);
  check_dev_f32(in, "in"); check_pm(out, "out");
  TORCH_CHECK(in.dim() == 4 && in.is_contiguous());
  const int B = in.size(0), C = in.size(1), H = in.size(2), W = in.size(3);
  TORCH_CHECK(out.size(0) == (int64_t)B * H * W && out.size(1) == C);
  check_ok(pfk_nchw_to_pm_f32(fptr(in), fptr(out), out.stride(0), 0, B, C, H, W, cur_stream()), "nchw_to_pm");
}

void pm_to_nchw(const Tensor& in, Tensor out) {
  OpScope scope(in);
  check_pm(in, "in"); check_dev_f32(out, "out");
  TORCH_CHECK(out.dim() == 4 && out.is_contiguous());
  const int B = out.size(0), C = out.size(1), H = out.size(2), W = out.size(3);
  TORCH_CHECK(in.size(0) == (int64_t)B * H * W && in.size(1) == C);
  check_ok(pfk_pm_to_nchw_f32(fptr(in), in.stride(0), 0, fptr(out), B, C, H, W, cur_stream()), "pm_to_nchw");
}

// in [B*N, C] pixel-major view -> out [B, C, Npad] channel-major (row stride = out.stride(1))
void pm_to_cm(const Tensor& in, Tensor out) {
  OpScope scope(in);
  check_pm(in, "in"); check_dev_f32(out, "out");
  TORCH_CHECK(out.dim() == 3 && out.stride(2) == 1 && out.stride(0) == out.size(1) * out.stride(1), "pm_to_cm: out [B,C,Npad]");
  const int B = out.size(0), C = out.size(1);
  TORCH_CHECK(in.size(1) == C && in.size(0) % B == 0);
  const int N = in.size(0) / B;
  TORCH_CHECK(out.stride(1) >= N);
  check_ok(pfk_pm_to_cm_f32(fptr(in), in.stride(0), fptr(out), out.stride(1), B, C, N, cur_stream()), "pm_to_cm");
}

void conv_stem(const Tensor& img, const Tensor& weight, const c10::optional<Tensor>& bias, Tensor out, bool relu) {
  OpScope scope(img);
  check_dev_f32(img, "img"); check_dev_f32(weight, "weight");
  const bool out_b16 = check_pm_any(out, "out");
  TORCH_CHECK(img.dim() == 4 && img.size(1) == 3 && img.is_contiguous(), "conv_stem: img [B,3,H,W] contiguous");
  const int B = img.size(0), H = img.size(2), W = img.size(3), cout = out.size(1);
  TORCH_CHECK(weight.is_contiguous() && weight.numel() == 49 * 3 * cout, "conv_stem: weight [49,3,cout]");
  TORCH_CHECK(out.size(0) == (int64_t)B * ((H - 1) / 2 + 1) * ((W - 1) / 2 + 1), "conv_stem: out rows");
  const float* b = nullptr;
  if (bias.has_value()) { check_dev_f32(*bias, "bias"); TORCH_CHECK(bias->numel() == cout && bias->is_contiguous()); b = fptr(*bias); }
  if (out_b16) {
    check_ok(pfk_conv_stem_b16(fptr(img), fptr(weight), b, out.data_ptr(), out.stride(0), B, H, W, cout, relu, cur_stream()), "conv_stem (bf16 out)");
    return;
  }
  check_ok(pfk_conv_stem_f32(fptr(img), fptr(weight), b, fptr(out), out.stride(0), B, H, W, cout, relu, cur_stream()), "conv_stem");
}

// weight / bias gradient in PyTorch's layouts: dw [cout, sum(reals), kh, kw], db [cout] (optional); dy [B*Ho*Wo, cout4]
void conv_wgrad_unpacked(at::TensorList srcs, const Tensor& dy, int64_t B, int64_t H, int64_t W, int64_t kh, int64_t kw, Tensor dw,
                         const c10::optional<Tensor>& db, at::IntArrayRef reals, int64_t stride, bool accumulate) {
  OpScope scope(dy);
  TORCH_CHECK(srcs.size() >= 1 && srcs.size() <= 3 && reals.size() == srcs.size(), "conv_wgrad_unpacked: 1..3 sources, one real count each");
  check_pm(dy, "dy"); check_dev_f32(dw, "dw");
  pfk_conv_desc d{};
  const int64_t M = B * H * W;
  int real[3] = {0, 0, 0};
  int64_t cin = 0;
  for (size_t i = 0; i < srcs.size(); ++i) {
    check_pm(srcs[i], "src");
    TORCH_CHECK(srcs[i].size(0) == M, "conv_wgrad_unpacked: src rows != B*H*W");
    d.src[i].ptr = fptr(srcs[i]); d.src[i].ld = srcs[i].stride(0); d.src[i].channels = srcs[i].size(1);
    real[i] = (int)reals[i];
    cin += reals[i];
  }
  d.num_src = srcs.size();
  d.B = B; d.H = H; d.W = W; d.kh = kh; d.kw = kw; d.cout = dy.size(1);
  TORCH_CHECK(stride >= 1, "conv_wgrad_unpacked: stride");
  d.stride = (int)stride;
  TORCH_CHECK(dy.size(0) == B * ((H - 1) / stride + 1) * ((W - 1) / stride + 1), "conv_wgrad_unpacked: dy rows != B*Ho*Wo");
  TORCH_CHECK(dw.is_contiguous() && dw.dim() == 4 && dw.size(0) <= d.cout && dw.size(1) == cin && dw.size(2) == kh && dw.size(3) == kw,
              "conv_wgrad_unpacked: dw [cout, sum(reals), kh, kw] contiguous");
  float* dbp = nullptr;
  if (db.has_value()) {
    check_dev_f32(*db, "db");
    TORCH_CHECK(db->is_contiguous() && db->numel() == dw.size(0), "conv_wgrad_unpacked: db [cout]");
    dbp = fptr(*db);
  }
  const long long need = pfk_conv_wgrad_unpacked_workspace_bytes(&d, dbp != nullptr);
  Tensor ws = at::empty({(int64_t)need}, dy.options().dtype(at::kByte));
  check_ok(pfk_conv_wgrad_unpacked_f32(&d, real, fptr(dy), dy.stride(0), (int)dw.size(0), fptr(dw), dbp, accumulate ? 1 : 0,
                                       ws.data_ptr(), need, cur_stream()), "conv_wgrad_unpacked");
}

int64_t instnorm_workspace_bytes(int64_t B, int64_t C) { return pfk_instnorm_workspace_bytes((int)B, (int)C); }

void instnorm_stats(const Tensor& x, int64_t B, int64_t HW, double eps, Tensor mean, Tensor rstd, Tensor workspace) {
  OpScope scope(x);
  const bool x_b16 = check_pm_any(x, "x");
  check_dev_f32(mean, "mean"); check_dev_f32(rstd, "rstd");
  const int C = x.size(1);
  TORCH_CHECK(x.size(0) == B * HW, "instnorm_stats: rows");
  TORCH_CHECK(mean.is_contiguous() && rstd.is_contiguous() && mean.numel() == B * C && rstd.numel() == B * C, "instnorm_stats: mean/rstd [B*C]");
  check_dev(workspace, "workspace");
  TORCH_CHECK(workspace.is_contiguous(), "instnorm_stats: workspace");
  if (x_b16) {
    check_ok(pfk_instnorm_stats_b16(x.data_ptr(), x.stride(0), (int)B, (int)HW, C, (float)eps, fptr(mean), fptr(rstd),
                                    workspace.data_ptr(), (long long)workspace.nbytes(), cur_stream()), "instnorm_stats (bf16 rows)");
    return;
  }
  check_ok(pfk_instnorm_stats_f32(fptr(x), x.stride(0), (int)B, (int)HW, C, (float)eps, fptr(mean), fptr(rstd),
                                  workspace.data_ptr(), (long long)workspace.nbytes(), cur_stream()), "instnorm_stats");
}

void norm_apply(const Tensor& x, const Tensor& mean, const Tensor& rstd, const c10::optional<Tensor>& residual, Tensor out,
                int64_t B, int64_t HW, bool relu, bool relu_after_residual) {
  OpScope scope(x);
  const bool x_b16 = check_pm_any(x, "x");
  check_dev_f32(mean, "mean"); check_dev_f32(rstd, "rstd");
  const bool out_b16 = check_pm_any(out, "out");
  TORCH_CHECK(out_b16 || !x_b16, "norm_apply: bf16 rows in need a bf16 output");
  const int C = x.size(1);
  TORCH_CHECK(x.size(0) == B * HW && out.size(0) == B * HW && out.size(1) == C, "norm_apply: shapes");
  TORCH_CHECK(mean.numel() == B * C && rstd.numel() == B * C, "norm_apply: mean/rstd [B*C]");
  if (out_b16) {      // the K8b encoders: bf16 residual rows (if any), bf16 output
    const void* r16 = nullptr; int r16_ld = 0;
    if (residual.has_value()) {
      check_pm_b16(*residual, "residual");
      TORCH_CHECK(residual->size(0) == B * HW && residual->size(1) == C, "norm_apply: residual shape");
      r16 = residual->data_ptr(); r16_ld = residual->stride(0);
    }
    check_ok(pfk_norm_apply_b16(x.data_ptr(), x_b16 ? 1 : 0, x.stride(0), fptr(mean), fptr(rstd), r16, r16_ld, out.data_ptr(), out.stride(0), (int)B, (int)HW, C,
                                relu, relu_after_residual, cur_stream()), "norm_apply (bf16 out)");
    return;
  }
  const float* r = nullptr; int r_ld = 0;
  if (residual.has_value()) {
    check_pm(*residual, "residual");
    TORCH_CHECK(residual->size(0) == B * HW && residual->size(1) == C, "norm_apply: residual shape");
    r = fptr(*residual); r_ld = residual->stride(0);
  }
  check_ok(pfk_norm_apply_f32(fptr(x), x.stride(0), fptr(mean), fptr(rstd), r, r_ld, fptr(out), out.stride(0), (int)B, (int)HW, C,
                              relu, relu_after_residual, cur_stream()), "norm_apply");
}

int64_t norm_bwd_workspace_bytes(int64_t B, int64_t C) { return pfk_norm_bwd_workspace_bytes((int)B, (int)C); }

// backward of relu?((x - mean) * rstd): dx, and the two per-(image, channel) sums (d beta / d gamma of an affine batch norm)
void norm_bwd(const Tensor& x, const Tensor& dy, const Tensor& mean, const Tensor& rstd, Tensor dx, Tensor sum_g, Tensor sum_gxhat,
              int64_t B, int64_t HW, bool relu) {
  OpScope scope(x);
  check_pm(x, "x"); check_pm(dy, "dy"); check_pm(dx, "dx"); check_dev_f32(mean, "mean"); check_dev_f32(rstd, "rstd");
  check_dev_f32(sum_g, "sum_g"); check_dev_f32(sum_gxhat, "sum_gxhat");
  const int C = x.size(1);
  TORCH_CHECK(x.size(0) == B * HW && dy.size(0) == B * HW && dx.size(0) == B * HW && dy.size(1) == C && dx.size(1) == C, "norm_bwd: shapes");
  TORCH_CHECK(mean.numel() == B * C && rstd.numel() == B * C && sum_g.numel() == B * C && sum_gxhat.numel() == B * C &&
              mean.is_contiguous() && rstd.is_contiguous() && sum_g.is_contiguous() && sum_gxhat.is_contiguous(), "norm_bwd: [B*C] statistics");
  const long long need = pfk_norm_bwd_workspace_bytes((int)B, C);
  Tensor ws = at::empty({(int64_t)need}, x.options().dtype(at::kByte));
  check_ok(pfk_norm_bwd_f32(fptr(x), x.stride(0), fptr(dy), dy.stride(0), fptr(mean), fptr(rstd), fptr(dx), dx.stride(0), fptr(sum_g),
                            fptr(sum_gxhat), (int)B, (int)HW, C, relu, ws.data_ptr(), need, cur_stream()), "norm_bwd");
}

// stem weight / bias gradient: img [B,3,H,W], dy pixel-major [B*Ho*Wo, cout] -> dw [49,3,cout], db [cout]
void conv_stem_wgrad(const Tensor& img, const Tensor& dy, Tensor dw, Tensor db) {
  OpScope scope(img);
  check_dev_f32(img, "img"); check_pm(dy, "dy"); check_dev_f32(dw, "dw"); check_dev_f32(db, "db");
  TORCH_CHECK(img.dim() == 4 && img.size(1) == 3 && img.is_contiguous(), "conv_stem_wgrad: img [B,3,H,W] contiguous");
  const int B = img.size(0), H = img.size(2), W = img.size(3), cout = dy.size(1);
  TORCH_CHECK(dy.size(0) == (int64_t)B * ((H - 1) / 2 + 1) * ((W - 1) / 2 + 1), "conv_stem_wgrad: dy rows");
  TORCH_CHECK(dw.is_contiguous() && dw.numel() == 49 * 3 * cout && db.is_contiguous() && db.numel() == cout, "conv_stem_wgrad: dw [49,3,cout], db [cout]");
  const long long need = pfk_conv_stem_wgrad_workspace_bytes();
  Tensor ws = at::empty({(int64_t)need}, img.options().dtype(at::kByte));
  check_ok(pfk_conv_stem_wgrad_f32(fptr(img), fptr(dy), dy.stride(0), fptr(dw), fptr(db), B, H, W, cout, ws.data_ptr(), need, cur_stream()),
           "conv_stem_wgrad");
}

void softmax_rows(Tensor x) {
  OpScope scope(x);
  check_dev_f32(x, "x");
  TORCH_CHECK(x.dim() >= 2 && x.is_contiguous(), "softmax_rows: contiguous [..., rows, cols]");
  const int64_t cols = x.size(-1), rows = x.numel() / cols;
  check_ok(pfk_softmax_rows_f32(fptr(x), rows, (int)cols, cols, cur_stream()), "softmax_rows");
}

void forward_interpolate(const Tensor& flow, Tensor out) {
  OpScope scope(flow);
  check_dev_f32(flow, "flow"); check_dev_f32(out, "out");
  TORCH_CHECK(flow.dim() == 4 && flow.size(1) == 2 && flow.is_contiguous() && out.is_contiguous() && out.sizes() == flow.sizes(),
              "forward_interpolate: flow/out [B,2,H,W] contiguous");
  check_ok(pfk_forward_interpolate_f32(fptr(flow), fptr(out), flow.size(0), flow.size(2), flow.size(3), cur_stream()), "forward_interpolate");
}

// weight gradient in the packed [cout, ktot] layout; dy [M, cout] (cout % 4 == 0)
void conv_wgrad(at::TensorList srcs, const Tensor& dy, int64_t B, int64_t H, int64_t W, int64_t kh, int64_t kw, Tensor out,
                bool with_bias, int64_t stride) {
  OpScope scope(dy);
  TORCH_CHECK(srcs.size() >= 1 && srcs.size() <= 3, "conv_wgrad: 1..3 sources");
  check_pm(dy, "dy"); check_dev_f32(out, "out");
  pfk_conv_desc d{};
  const int64_t M = B * H * W;
  for (size_t i = 0; i < srcs.size(); ++i) {
    check_pm(srcs[i], "src");
    TORCH_CHECK(srcs[i].size(0) == M, "conv_wgrad: src rows != B*H*W");
    d.src[i].ptr = fptr(srcs[i]); d.src[i].ld = srcs[i].stride(0); d.src[i].channels = srcs[i].size(1);
  }
  d.num_src = srcs.size();
  d.B = B; d.H = H; d.W = W; d.kh = kh; d.kw = kw; d.cout = dy.size(1);
  TORCH_CHECK(stride >= 1, "conv_wgrad: stride");
  d.stride = (int)stride;
  TORCH_CHECK(dy.size(0) == B * ((H - 1) / stride + 1) * ((W - 1) / stride + 1), "conv_wgrad: dy rows != B*Ho*Wo");
  TORCH_CHECK(out.is_contiguous() && out.dim() == 2 && out.size(0) == d.cout && out.size(1) == pfk_conv_ktot(&d) + (with_bias ? 32 : 0),
              "conv_wgrad: out [cout, ktot (+32 with the bias column)]");
  const long long need = pfk_conv_wgrad_workspace_bytes(&d, with_bias);
  Tensor ws;
  void* wsp = nullptr;
  if (need > 0) { ws = at::empty({(int64_t)need}, dy.options().dtype(at::kByte)); wsp = ws.data_ptr(); }
  check_ok(pfk_conv_wgrad_f32(&d, fptr(dy), dy.stride(0), fptr(out), with_bias, wsp, need, cur_stream()), "conv_wgrad");
}

void gru_gates_zr(const Tensor& a_zr, const Tensor& h, Tensor z, Tensor r, Tensor rh) {
  OpScope scope(h);
  check_pm(h, "h"); check_dev_f32(a_zr, "a_zr");
  const int64_t M = h.size(0); const int C = h.size(1);
  TORCH_CHECK(a_zr.is_contiguous() && a_zr.size(0) == M && a_zr.size(1) == 2 * C && z.is_contiguous() && r.is_contiguous() && rh.is_contiguous());
  check_ok(pfk_gru_gates_zr_f32(fptr(a_zr), fptr(h), h.stride(0), fptr(z), fptr(r), fptr(rh), M, C, cur_stream()), "gru_gates_zr");
}
void gru_gates_q(const Tensor& a_q, const Tensor& z, const Tensor& h, Tensor q, Tensor h_new) {
  OpScope scope(h);
  check_pm(h, "h");
  const int64_t M = h.size(0); const int C = h.size(1);
  TORCH_CHECK(a_q.is_contiguous() && a_q.size(0) == M && a_q.size(1) == C && z.is_contiguous() && q.is_contiguous() && h_new.is_contiguous());
  check_ok(pfk_gru_gates_q_f32(fptr(a_q), fptr(z), fptr(h), h.stride(0), fptr(q), fptr(h_new), M, C, cur_stream()), "gru_gates_q");
}
void gru_backward_q(const Tensor& dh_new, const Tensor& z, const Tensor& q, const Tensor& h, Tensor da_q, Tensor da_zr, Tensor dh) {
  OpScope scope(h);
  check_pm(h, "h"); check_pm(dh_new, "dh_new");
  const int64_t M = h.size(0); const int C = h.size(1);
  TORCH_CHECK(z.is_contiguous() && q.is_contiguous() && da_q.is_contiguous() && da_zr.is_contiguous() && dh.is_contiguous() &&
              da_zr.size(1) == 2 * C && dh_new.size(1) == C);
  check_ok(pfk_gru_backward_q_f32(fptr(dh_new), dh_new.stride(0), fptr(z), fptr(q), fptr(h), h.stride(0), fptr(da_q), fptr(da_zr),
                                  fptr(dh), M, C, cur_stream()), "gru_backward_q");
}
void gru_backward_zr(const Tensor& d_rh, const Tensor& h, const Tensor& r, Tensor da_zr, Tensor dh) {
  OpScope scope(h);
  check_pm(h, "h");
  const int64_t M = h.size(0); const int C = h.size(1);
  TORCH_CHECK(d_rh.is_contiguous() && r.is_contiguous() && da_zr.is_contiguous() && dh.is_contiguous() && da_zr.size(1) == 2 * C);
  check_ok(pfk_gru_backward_zr_f32(fptr(d_rh), fptr(h), h.stride(0), fptr(r), fptr(da_zr), fptr(dh), M, C, cur_stream()), "gru_backward_zr");
}

int64_t abi_version() { return pfk_abi_version(); }
int64_t conv_workspace_bytes() { return pfk_conv_workspace_bytes(); }
int64_t conv_workspace_fault_offset() { return pfk_conv_workspace_fault_offset(); }
// inert (RuntimeError) unless the process opted in with PFK_DEBUG_KNOBS=1: process-global kernel selection (include/pfk.h)
void debug_set_tile(int64_t cfg) { check_ok(pfk_debug_set_tile((int)cfg), "debug_set_tile"); }
void debug_set_lookup_pix(int64_t pix) { check_ok(pfk_debug_set_lookup_pix((int)pix), "debug_set_lookup_pix"); }
void debug_set_wgrad(int64_t v) { check_ok(pfk_debug_set_wgrad((int)v), "debug_set_wgrad"); }
void debug_set_altcorr(int64_t v) { check_ok(pfk_debug_set_altcorr((int)v), "debug_set_altcorr"); }
#ifndef PFK_SOURCE_HASH
#define PFK_SOURCE_HASH "unstamped"
#endif
// the same stamp as a record that ptlflow_amd/_build.py finds in the FILE (embedded_hash): staleness is decided without loading it
__attribute__((used)) static const char pfk_ext_stamp_record[] = "PFK_EXT_SOURCE_HASH=" PFK_SOURCE_HASH;
std::string source_hash() { return std::string(pfk_ext_stamp_record + 20) + ":" + pfk_source_hash(); }   // "<this extension's stamp>:<libpfk.so's stamp>"

}  // namespace

TORCH_LIBRARY(pfk, m) {
  m.def("abi_version() -> int", &abi_version);
  m.def("source_hash() -> str", &source_hash);
  m.def("gru_gates_zr(Tensor a_zr, Tensor h, Tensor(a!) z, Tensor(b!) r, Tensor(c!) rh) -> ()");
  m.def("gru_gates_q(Tensor a_q, Tensor z, Tensor h, Tensor(a!) q, Tensor(b!) h_new) -> ()");
  m.def("gru_backward_q(Tensor dh_new, Tensor z, Tensor q, Tensor h, Tensor(a!) da_q, Tensor(b!) da_zr, Tensor(c!) dh) -> ()");
  m.def("gru_backward_zr(Tensor d_rh, Tensor h, Tensor r, Tensor(a!) da_zr, Tensor(b!) dh) -> ()");
  m.def("conv_wgrad(Tensor[] srcs, Tensor dy, int B, int H, int W, int kh, int kw, Tensor(a!) out, bool with_bias=False, int stride=1) -> ()");
  m.def("conv_wgrad_unpacked(Tensor[] srcs, Tensor dy, int B, int H, int W, int kh, int kw, Tensor(a!) dw, Tensor(b!)? db, int[] reals, int stride=1, bool accumulate=False) -> ()");
  m.def("forward_interpolate(Tensor flow, Tensor(a!) out) -> ()");
  m.def("softmax_rows(Tensor(a!) x) -> ()");
  m.def("norm_bwd(Tensor x, Tensor dy, Tensor mean, Tensor rstd, Tensor(a!) dx, Tensor(b!) sum_g, Tensor(c!) sum_gxhat, int B, int HW, bool relu) -> ()");
  m.def("conv_stem_wgrad(Tensor img, Tensor dy, Tensor(a!) dw, Tensor(b!) db) -> ()");
  m.def("instnorm_workspace_bytes(int B, int C) -> int", &instnorm_workspace_bytes);
  m.def("conv_stem(Tensor img, Tensor weight, Tensor? bias, Tensor(a!) out, bool relu) -> ()");
  m.def("instnorm_stats(Tensor x, int B, int HW, float eps, Tensor(a!) mean, Tensor(b!) rstd, Tensor(c!) workspace) -> ()");
  m.def("norm_apply(Tensor x, Tensor mean, Tensor rstd, Tensor? residual, Tensor(a!) out, int B, int HW, bool relu, "
        "bool relu_after_residual) -> ()");
  m.def("debug_set_tile(int cfg) -> ()", &debug_set_tile);
  m.def("debug_set_lookup_pix(int pix) -> ()", &debug_set_lookup_pix);
  m.def("debug_set_wgrad(int variant) -> ()", &debug_set_wgrad);
  m.def("debug_set_altcorr(int mode) -> ()", &debug_set_altcorr);
  m.def("corr_volume(Tensor f1, Tensor f2, float scale, Tensor(a!) out) -> ()");
  m.def("corr_pool2x2(Tensor inp, Tensor(a!) out) -> ()");
  m.def("corr_volume_bf16(Tensor f1, Tensor f2, float scale, Tensor(a!) out) -> ()");
  m.def("fmap_pool2x2(Tensor inp, Tensor(a!) out, int B, int H, int W) -> ()");
  m.def("corr_lookup(Tensor[] levels, Tensor coords, int radius, Tensor(a!) out) -> ()");
  m.def("fmap_to_blocked(Tensor inp, Tensor(a!) out, int B, int H, int W) -> ()");
  m.def("corr_pool2x2_blocked(Tensor inp, Tensor(a!) out, int H, int W) -> ()");
  m.def("corr_lookup_blocked(Tensor[] levels, int[] lvl_h, int[] lvl_w, Tensor coords, int radius, Tensor(a!) out) -> ()");
  m.def("blocked_map_elems(int H, int W) -> int", &blocked_map_elems);
  m.def("conv2d(Tensor[] srcs, int B, int H, int W, int kh, int kw, Tensor weight, Tensor? bias, int cout, "
        "int epilogue, bool relu, float scale, Tensor(a!)? out, Tensor(b!)? h, Tensor(c!)? aux_z, Tensor(d!)? aux_rh, "
        "Tensor(e!)? workspace=None, Tensor? residual=None, int stride=1, bool relu_after_residual=False, int cout_active=0, int cout_split=0) -> ()");
  m.def("conv2d_group(Tensor[] srcs, int B, int H, int W, int[] k, Tensor[] weights, Tensor[] biases, int[] relu, float[] scale, Tensor(a!)[] outs, Tensor[] residuals) -> ()");
  m.def("conv2d_b16(Tensor[] srcs, int B, int H, int W, int kh, int kw, Tensor weight, Tensor? bias, int cout, int epilogue, bool relu, "
        "float scale, Tensor(a!)? out, Tensor(b!)? h=None, Tensor(c!)? h_b16=None, Tensor(d!)? aux_z=None, Tensor(e!)? aux_rh=None, "
        "Tensor? residual=None, int stride=1, bool relu_after_residual=False) -> ()");
  m.def("debug_set_b16(int cfg) -> ()", &debug_set_b16);
  m.def("debug_set_stem_valu(int on) -> ()", &debug_set_stem_valu);
  m.def("debug_set_cin2_valu(int on) -> ()", &debug_set_cin2_valu);
  m.def("conv_workspace_bytes() -> int", &conv_workspace_bytes);
  m.def("conv_workspace_fault_offset() -> int", &conv_workspace_fault_offset);
  m.def("conv_cin2(Tensor inp, Tensor weight, Tensor? bias, Tensor(a!) out, int B, int H, int W, int k, bool relu) -> ()");
  m.def("flow_delta(Tensor inp, Tensor weight, Tensor? bias, Tensor coords0, Tensor(a!) coords1, Tensor(b!)? delta_out, "
        "Tensor(c!)? flow_out, Tensor(d!)? flow_out_b16=None) -> ()");
  m.def("flow_from_coords(Tensor coords0, Tensor coords1, Tensor(a!) flow_out) -> ()");
  m.def("convex_upsample(Tensor flow, Tensor mask, Tensor(a!) out) -> ()");
  m.def("convex_upsample_pm(Tensor flow_pm, Tensor mask, Tensor(a!) out) -> ()");
  m.def("mask_upsample(Tensor x, Tensor weight_perm, Tensor? bias_perm, float scale, Tensor flow_pm, Tensor(a!) out) -> ()");
  m.def("upflow8(Tensor coords0, Tensor coords1, Tensor(a!) out) -> ()");
  m.def("altcorr_forward(Tensor fmap1, Tensor fmap2, Tensor coords, int radius) -> Tensor");
  m.def("altcorr_backward(Tensor fmap1, Tensor fmap2, Tensor coords, Tensor corr_grad, int radius) -> Tensor[]");
  m.def("corr_lookup_bwd(Tensor(a!)[] grad_levels, int[] lvl_h, int[] lvl_w, Tensor coords, int radius, Tensor grad_out) -> ()");
  m.def("corr_volume_bwd(Tensor dC, int N2, Tensor f1, Tensor f2cm, float scale, Tensor(a!) df1, bool accumulate, Tensor(b!) df2) -> ()");
  m.def("convex_upsample_bwd(Tensor flow, Tensor mask, Tensor grad_out, Tensor(a!) grad_mask, Tensor(b!) grad_flow) -> ()");
  m.def("nchw_to_pm(Tensor inp, Tensor(a!) out) -> ()");
  m.def("pm_to_nchw(Tensor inp, Tensor(a!) out) -> ()");
  m.def("pm_to_cm(Tensor inp, Tensor(a!) out) -> ()");
}

TORCH_LIBRARY_IMPL(pfk, CUDA, m) {
  m.impl("corr_volume", &corr_volume);
  m.impl("corr_pool2x2", &corr_pool2x2);
  m.impl("corr_volume_bf16", &corr_volume_bf16);
  m.impl("fmap_pool2x2", &fmap_pool2x2);
  m.impl("corr_lookup", &corr_lookup);
  m.impl("fmap_to_blocked", &fmap_to_blocked);
  m.impl("corr_pool2x2_blocked", &corr_pool2x2_blocked);
  m.impl("corr_lookup_blocked", &corr_lookup_blocked);
  m.impl("conv2d", &conv2d);
  m.impl("conv2d_b16", &conv2d_b16);
  m.impl("conv2d_group", &conv2d_group);
  m.impl("conv_cin2", &conv_cin2);
  m.impl("flow_delta", &flow_delta);
  m.impl("flow_from_coords", &flow_from_coords);
  m.impl("convex_upsample", &convex_upsample);
  m.impl("convex_upsample_pm", &convex_upsample_pm);
  m.impl("mask_upsample", &mask_upsample);
  m.impl("upflow8", &upflow8);
  m.impl("altcorr_forward", &altcorr_forward);
  m.impl("altcorr_backward", &altcorr_backward);
  m.impl("corr_lookup_bwd", &corr_lookup_bwd);
  m.impl("corr_volume_bwd", &corr_volume_bwd);
  m.impl("convex_upsample_bwd", &convex_upsample_bwd);
  m.impl("nchw_to_pm", &nchw_to_pm);
  m.impl("pm_to_nchw", &pm_to_nchw);
  m.impl("pm_to_cm", &pm_to_cm);
  m.impl("conv_stem", &conv_stem);
  m.impl("gru_gates_zr", &gru_gates_zr);
  m.impl("gru_gates_q", &gru_gates_q);
  m.impl("gru_backward_q", &gru_backward_q);
  m.impl("gru_backward_zr", &gru_backward_zr);
  m.impl("conv_wgrad", &conv_wgrad);
  m.impl("conv_wgrad_unpacked", &conv_wgrad_unpacked);
  m.impl("forward_interpolate", &forward_interpolate);
  m.impl("softmax_rows", &softmax_rows);
  m.impl("norm_bwd", &norm_bwd);
  m.impl("conv_stem_wgrad", &conv_stem_wgrad);
  m.impl("instnorm_stats", &instnorm_stats);
  m.impl("norm_apply", &norm_apply);
}
