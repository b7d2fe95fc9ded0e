// kernels_optim.hip -- tcnn Trainer::optimizer_step for Ema{ ExponentialDecay{ Adam } }
// (call site CORE/src/nerf_model.cu:1644,1681; hyper-parameters CORE/configs/base.json:5-22).
// One fused pass over the flat parameter vector: gradient read + reset (replaces the Overwrite
// memset of tcnn's backward), Adam on fp32 master weights, fp16 working copy, EMA shadow copy.
// SURVEY TCNN-A6/A7/A8: grid entries with zero gradient are skipped by Adam (not by the EMA),
// L2 regularisation only on the MLP matrices, per-parameter step counters, debiased EMA.
// The last block to finish advances the device-resident step / iteration counters and applies
// the exponential LR decay, so a whole training run needs no host synchronisation.
#include "device_common.h"
#include "model.h"

namespace mon {

__global__ void __launch_bounds__(256) k_optimizer(ParamPtrs p, OptimConst oc, DevState* __restrict__ st) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t n_valid = st->n_valid, step = st->step;
    const float lr0 = st->lr;
    __shared__ float s_deb[2];
    if (threadIdx.x == 0) {     // EMA debias factors from the global step after increment (double pow once per block)
        const uint32_t cur = step + 1u;
        s_deb[0] = 1.f - (float)pow((double)oc.ema_decay, (double)(cur - 1u));
        s_deb[1] = 1.f / (1.f - (float)pow((double)oc.ema_decay, (double)cur));
    }
    __syncthreads();
    if (n_valid != 0u && i < oc.n_params) {
        const bool is_matrix = i < oc.n_mlp;
        float g;
        if (is_matrix) { g = p.gmlp[i] / oc.loss_scale; p.gmlp[i] = 0.f; }
        else {
            half_t* gp = reinterpret_cast<half_t*>(p.ggrid) + (i - oc.n_mlp);
            const half_t gh = *gp;
            g = (float)gh / oc.loss_scale;
            if ((float)gh != 0.f) *gp = (half_t)0.f;
        }
        float w_half;
        if (is_matrix || g != 0.f) {
            const float w = p.master[i];
            if (is_matrix) g += oc.l2_reg * w;
            const float gsq = g * g;
            const float fm = oc.beta1 * p.m1[i] + (1.f - oc.beta1) * g;
            const float sm = oc.beta2 * p.m2[i] + (1.f - oc.beta2) * gsq;
            p.m1[i] = fm; p.m2[i] = sm;
            const uint32_t cs = p.steps[i] + 1u; p.steps[i] = cs;
            const float lr = lr0 * sqrtf(1.f - powf(oc.beta2, (float)cs)) / (1.f - powf(oc.beta1, (float)cs));
            const float eff = lr / (sqrtf(sm) + oc.epsilon);
            const float nw = w - eff * fm;
            p.master[i] = nw;
            const half_t nh = (half_t)nw;
            reinterpret_cast<half_t*>(p.half)[i] = nh;
            w_half = (float)nh;
        } else {
            w_half = (float)reinterpret_cast<const half_t*>(p.half)[i];
        }
        // EMA (ema_step_half_precision)
        const float d = oc.ema_decay;
        const float deb_old = s_deb[0], deb_new = s_deb[1];
        half_t* ep = reinterpret_cast<half_t*>(p.ema) + i;
        *ep = (half_t)((((float)*ep * d) * deb_old + w_half * (1.f - d)) * deb_new);
    }
    // ---- last block advances the counters
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        const uint32_t t = atomicAdd(&st->ticket, 1u);
        if (t == gridDim.x - 1u) {
            st->ticket = 0u;
            st->iter = st->iter + 1u;
            if (n_valid != 0u) {
                const uint32_t cur = step + 1u;
                st->step = cur;
                if ((int)cur >= oc.decay_start && oc.decay_interval > 0 && ((int)cur - oc.decay_start) % oc.decay_interval == 0) st->lr = lr0 * oc.decay_base;
            } else {
                st->skipped = st->skipped + 1u;
            }
            __threadfence();
        }
    }
}

void launch_optimizer(hipStream_t s, const ParamPtrs& p, const OptimConst& oc, DevState* st) {
    hipLaunchKernelGGL(k_optimizer, dim3((oc.n_params + 255) / 256), dim3(256), 0, s, p, oc, st);
}

}  // namespace mon
