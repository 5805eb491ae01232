// Fused multi-tensor AdamW step for the encoder / OcOccNet parameter set.
//
// The reference trains with torch.optim.AdamW through mmcv's optimizer hook
// (configs/_base_/schedules/cosine_2x.py:2-8: optimizer = dict(type='AdamW', betas=(0.9, 0.999),
// weight_decay=0.05, norm layers decay_mult 0); lr override at configs/ococc/ococcnet.py:468-470).  On this workload the parameters are ~3e5 floats in 9 tensors; the stock
// multi-tensor kernel walks them in 64K-element chunks (a handful of workgroups, ~40 us).  Here one
// launch covers every tensor with one thread per 4 elements, and the step counter lives on the
// device so that the launch can sit inside a captured HIP graph.
#include "common.hpp"
#include "weight_layout.hpp"

namespace {

constexpr int kMaxTensors = 48;
constexpr int kElemsPerBlock = 256 * 4;
constexpr int kTicketMaxBlocks = 2048;

struct AdamPack {
  float* p[kMaxTensors];
  const float* g[kMaxTensors];
  float* m[kMaxTensors];
  float* v[kMaxTensors];
  int64_t n[kMaxTensors];
  int32_t first_block[kMaxTensors + 1];
  int32_t count;
};

// bf16 operand layouts of convolution weights refreshed by the update itself (ococc_adamw_operands_f32): the separate
// ococc_weight_prepare_multi_bf16 launch at the start of the next step reads the parameters the optimizer has just written
constexpr int kMaxOperands = 8;
struct OperandPack {
  uint16_t* dst[kMaxOperands];
  int32_t tensor[kMaxOperands], mode[kMaxOperands], kvol[kMaxOperands], cin[kMaxOperands], cout[kMaxOperands];
  int32_t count;
};

__global__ void __launch_bounds__(256)
adamw_kernel(AdamPack pk, float lr_arg, const float* __restrict__ lr_dev, float beta1, float beta2, float eps, float wd,
             float* step, uint32_t* ticket, OperandPack ops) {
  // the learning rate either rides in the launch arguments or is read from device memory: a schedule can then
  // change it between replays of a captured HIP graph (a kernel argument is frozen into the graph node)
  const float lr = lr_dev ? *lr_dev : lr_arg;
  // The step count is bumped either by a one-thread kernel queued behind this one, or -- `ticket` given, small
  // launches only -- by the LAST workgroup to finish: every workgroup has read the old count by the time it
  // takes its ticket, so the store cannot be seen by this launch.  (One same-address atomic per workgroup:
  // nothing for the 300 workgroups of the encoder, ~0.5 ms for the 66 M parameters of OcOccNet, hence the choice.)
  const float t = *(const volatile float*)step + 1.f;
  // binary search of this block's tensor (the table sits in kernel-argument memory: every probe is a
  // dependent scalar load, so a linear walk over 48 entries costs microseconds per block)
  int lo = 0, hi = pk.count - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if ((int)blockIdx.x >= pk.first_block[mid]) lo = mid; else hi = mid - 1;
  }
  const int ti = lo;
  const int64_t base = (int64_t)(blockIdx.x - pk.first_block[ti]) * kElemsPerBlock + threadIdx.x * 4;
  const int64_t n = pk.n[ti];
  // bias corrections once per workgroup (two powf per thread would cost more than the update itself)
  __shared__ float bc[2];
  if (threadIdx.x == 0) {
    bc[0] = lr / (1.f - powf(beta1, t));
    bc[1] = 1.f / sqrtf(1.f - powf(beta2, t));
  }
  __syncthreads();
  const float step_size = bc[0];
  const float inv_sqrt_bc2 = bc[1];
  float* __restrict__ p = pk.p[ti];
  const float* __restrict__ g = pk.g[ti];
  float* __restrict__ m = pk.m[ti];
  float* __restrict__ v = pk.v[ti];
  if (base + 3 < n && (((uintptr_t)(p + base) | (uintptr_t)(g + base) | (uintptr_t)(m + base) |
                        (uintptr_t)(v + base)) & 15) == 0) {
    f32x4 pv = *(f32x4*)(p + base), mv = *(f32x4*)(m + base), vv = *(f32x4*)(v + base);
    const f32x4 gv = *(const f32x4*)(g + base);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float pj = pv[j] * (1.f - lr * wd);
      const float mj = beta1 * mv[j] + (1.f - beta1) * gv[j];
      const float vj = beta2 * vv[j] + (1.f - beta2) * gv[j] * gv[j];
      pj -= step_size * (mj / (sqrtf(vj) * inv_sqrt_bc2 + eps));
      pv[j] = pj; mv[j] = mj; vv[j] = vj;
    }
    *(f32x4*)(p + base) = pv;
    *(f32x4*)(m + base) = mv;
    *(f32x4*)(v + base) = vv;
    for (int o = 0; o < ops.count; ++o)   // (workgroup-uniform: at most a few entries, none for most tensors)
      if (ops.tensor[o] == ti) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          ops.dst[o][ococc_operand_index(ops.mode[o], ops.kvol[o], ops.cin[o], ops.cout[o], base + j)] = ococc_f32_to_bf16(pv[j]);
      }
  } else {
    for (int64_t i = base; i < base + 4 && i < n; ++i) {
      const float gj = g[i];
      float pj = p[i] * (1.f - lr * wd);
      const float mj = beta1 * m[i] + (1.f - beta1) * gj;
      const float vj = beta2 * v[i] + (1.f - beta2) * gj * gj;
      pj -= step_size * (mj / (sqrtf(vj) * inv_sqrt_bc2 + eps));
      p[i] = pj; m[i] = mj; v[i] = vj;
      for (int o = 0; o < ops.count; ++o)
        if (ops.tensor[o] == ti)
          ops.dst[o][ococc_operand_index(ops.mode[o], ops.kvol[o], ops.cin[o], ops.cout[o], i)] = ococc_f32_to_bf16(pj);
    }
  }
  if (ticket) {
    __syncthreads();  // (every thread of the workgroup is past its read of *step: it sits in front of the barrier above)
    if (threadIdx.x == 0) {
      const uint32_t got = atomicAdd(ticket, 1u);
      if (got == gridDim.x - 1) {
        *ticket = 0;  // ready for the next launch (stream order)
        *step = t;
      }
    }
  }
}

__global__ void adamw_bump_kernel(float* step) { *step += 1.f; }

}  // namespace

namespace {
int adamw_launch(int32_t num_tensors, void* const* params, const void* const* grads, void* const* exp_avg,
                 void* const* exp_avg_sq, const int64_t* numel, float lr, const float* lr_dev, float beta1, float beta2,
                 float eps, float weight_decay, float* step, int32_t bump_step, hipStream_t stream,
                 const OperandPack* operands = nullptr, const int32_t* operand_tensor = nullptr) {
  OCOCC_REQUIRE(num_tensors >= 0, "negative tensor count");
  OCOCC_REQUIRE(step, "step must be a device pointer");
  OCOCC_REQUIRE(bump_step >= 0 && bump_step <= 2, "bump_step must be 0, 1 or 2");
  if (num_tensors == 0) return OCOCC_OK;
  OCOCC_REQUIRE(params && grads && exp_avg && exp_avg_sq && numel, "null pointer table");
  OCOCC_REQUIRE(num_tensors <= kMaxTensors, "at most 48 tensors per call (split the parameter list)");
  AdamPack pk;
  OperandPack ops;
  ops.count = 0;
  if (operands) {
    ops = *operands;
    for (int o = 0; o < ops.count; ++o) ops.tensor[o] = -1;
  }
  int blocks = 0, cnt = 0;
  for (int i = 0; i < num_tensors; ++i) {
    OCOCC_REQUIRE(numel[i] >= 0, "negative numel");
    if (numel[i] == 0) continue;
    OCOCC_REQUIRE(params[i] && grads[i] && exp_avg[i] && exp_avg_sq[i], "null tensor pointer");
    pk.p[cnt] = (float*)params[i];
    pk.g[cnt] = (const float*)grads[i];
    pk.m[cnt] = (float*)exp_avg[i];
    pk.v[cnt] = (float*)exp_avg_sq[i];
    pk.n[cnt] = numel[i];
    if (operands)   // (tensor numbers of the operand entries follow the compaction of empty tensors)
      for (int o = 0; o < operands->count; ++o)
        if (operand_tensor[o] == i) ops.tensor[o] = cnt;
    pk.first_block[cnt] = blocks;
    blocks += (int)ococc_cdiv(numel[i], kElemsPerBlock);
    ++cnt;
  }
  if (cnt == 0) return OCOCC_OK;
  pk.first_block[cnt] = blocks;
  pk.count = cnt;
  // bump_step 2: `step` points to {float count; uint32 ticket (zero)}; the last workgroup stores count + 1
  const bool in_kernel = bump_step == 2 && blocks <= kTicketMaxBlocks;
  hipLaunchKernelGGL(adamw_kernel, dim3(blocks), dim3(256), 0, stream, pk, lr, lr_dev, beta1, beta2, eps,
                     weight_decay, step, in_kernel ? (uint32_t*)(step + 1) : (uint32_t*)nullptr, ops);
  OCOCC_CHECK_LAUNCH();
  if (bump_step && !in_kernel) {
    hipLaunchKernelGGL(adamw_bump_kernel, dim3(1), dim3(1), 0, stream, step);
    OCOCC_CHECK_LAUNCH();
  }
  return OCOCC_OK;
}
}  // namespace

extern "C" int ococc_adamw_f32(int32_t num_tensors, void* const* params, const void* const* grads,
                               void* const* exp_avg, void* const* exp_avg_sq, const int64_t* numel,
                               float lr, float beta1, float beta2, float eps, float weight_decay,
                               float* step, int32_t bump_step, ococc_stream_t stream_) {
  return adamw_launch(num_tensors, params, grads, exp_avg, exp_avg_sq, numel, lr, nullptr, beta1, beta2, eps,
                      weight_decay, step, bump_step, (hipStream_t)stream_);
}

extern "C" int ococc_adamw_lr_dev_f32(int32_t num_tensors, void* const* params, const void* const* grads,
                                      void* const* exp_avg, void* const* exp_avg_sq, const int64_t* numel,
                                      const float* lr_dev, float beta1, float beta2, float eps, float weight_decay,
                                      float* step, int32_t bump_step, ococc_stream_t stream_) {
  OCOCC_REQUIRE(lr_dev, "lr_dev must be a device pointer");
  return adamw_launch(num_tensors, params, grads, exp_avg, exp_avg_sq, numel, 0.f, lr_dev, beta1, beta2, eps,
                      weight_decay, step, bump_step, (hipStream_t)stream_);
}

extern "C" int ococc_adamw_operands_f32(int32_t num_tensors, void* const* params, const void* const* grads,
                                        void* const* exp_avg, void* const* exp_avg_sq, const int64_t* numel, float lr,
                                        const float* lr_dev, float beta1, float beta2, float eps, float weight_decay,
                                        float* step, int32_t bump_step, int32_t num_operands,
                                        const int32_t* operand_tensor, const int32_t* operand_mode,
                                        const int32_t* operand_kvol, const int32_t* operand_cin,
                                        const int32_t* operand_cout, void* const* operand_dst, ococc_stream_t stream_) {
  OCOCC_REQUIRE(num_operands >= 0 && num_operands <= kMaxOperands, "at most 8 operand layouts per call");
  OperandPack ops;
  ops.count = num_operands;
  if (num_operands > 0) {
    OCOCC_REQUIRE(operand_tensor && operand_mode && operand_kvol && operand_cin && operand_cout && operand_dst,
                  "null operand table");
    for (int o = 0; o < num_operands; ++o) {
      OCOCC_REQUIRE(operand_tensor[o] >= 0 && operand_tensor[o] < num_tensors && operand_dst[o], "bad operand entry");
      OCOCC_REQUIRE((operand_mode[o] & 3) <= 2 && operand_mode[o] >= 0 && operand_mode[o] < 8, "bad operand mode");
      OCOCC_REQUIRE(operand_kvol[o] > 0 && operand_cin[o] > 0 && operand_cout[o] > 0 &&
                        (int64_t)operand_kvol[o] * operand_cin[o] * operand_cout[o] == numel[operand_tensor[o]],
                    "operand shape does not match the tensor");
      if (operand_mode[o] & 4) {
        const int kd = (operand_mode[o] & 3) == 0 ? operand_cin[o] : operand_cout[o];
        const int nc = (operand_mode[o] & 3) == 0 ? operand_cout[o] : operand_cin[o];
        OCOCC_REQUIRE(kd % 32 == 0 && nc % 16 == 0, "fragment-major layouts need kd % 32 == 0 and columns % 16 == 0");
      }
      ops.dst[o] = (uint16_t*)operand_dst[o];
      ops.mode[o] = operand_mode[o];
      ops.kvol[o] = operand_kvol[o];
      ops.cin[o] = operand_cin[o];
      ops.cout[o] = operand_cout[o];
    }
  }
  return adamw_launch(num_tensors, params, grads, exp_avg, exp_avg_sq, numel, lr, lr_dev, beta1, beta2, eps, weight_decay,
                      step, bump_step, (hipStream_t)stream_, &ops, operand_tensor);
}
