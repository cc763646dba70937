// Optimizer-side kernels of the train step (reference trainer.py:361-382): loss gradient, fused Adam + EMA on the
// flat parameter buffer (optax.adam semantics: bias-corrected, eps outside the sqrt, no weight decay).
#include "vdx_common.h"
#include "vdx_internal.h"

namespace vdx {

// d(mean loss)/d(eps_hat) in the UNet's channel-last layout; noise is [B,C,F,H,W]
__global__ __launch_bounds__(256) void loss_grad_kernel(const float* __restrict__ eps_hat, const float* __restrict__ noise,
                                                        float* __restrict__ d_eps, int B, int Cc, long fhw, int l2, float inv_count) {
    const long n = (long)B * Cc * fhw;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long b = i / (Cc * fhw), r = i - b * Cc * fhw;
        const long c = r / fhw, p = r - c * fhw;
        const long j = (b * fhw + p) * Cc + c;
        const float d = eps_hat[j] - noise[i];
        d_eps[j] = l2 ? 2.0f * d * inv_count : ((d > 0.f) - (d < 0.f)) * inv_count;
    }
}

// p, m, v, ema: flat fp32 [n]; g: gradient (already averaged over ranks).  grad_scale folds 1/world into the read.
__global__ __launch_bounds__(256) void adam_ema_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                       float* __restrict__ v, float* __restrict__ ema, long n, float lr, float b1, float b2,
                                                       float eps, float bc1, float bc2, float grad_scale, int do_ema, float decay) {
    for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (long)gridDim.x * blockDim.x * 4) {
        if (i + 3 < n) {
            float4 pv = *reinterpret_cast<float4*>(p + i), gv = *reinterpret_cast<const float4*>(g + i);
            float4 mv = *reinterpret_cast<float4*>(m + i), vv = *reinterpret_cast<float4*>(v + i);
            float* pp = &pv.x; float* gg = &gv.x; float* mm = &mv.x; float* v2 = &vv.x;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float gk = gg[k] * grad_scale;
                mm[k] = b1 * mm[k] + (1.f - b1) * gk;
                v2[k] = b2 * v2[k] + (1.f - b2) * gk * gk;
                pp[k] -= lr * (mm[k] / bc1) / (sqrtf(v2[k] / bc2) + eps);
            }
            *reinterpret_cast<float4*>(p + i) = pv; *reinterpret_cast<float4*>(m + i) = mv; *reinterpret_cast<float4*>(v + i) = vv;
            if (do_ema) {
                float4 e = *reinterpret_cast<float4*>(ema + i);
                e.x = decay * e.x + (1.f - decay) * pv.x; e.y = decay * e.y + (1.f - decay) * pv.y;
                e.z = decay * e.z + (1.f - decay) * pv.z; e.w = decay * e.w + (1.f - decay) * pv.w;
                *reinterpret_cast<float4*>(ema + i) = e;
            }
        } else {
            for (long k = i; k < n; ++k) {
                const float gk = g[k] * grad_scale;
                m[k] = b1 * m[k] + (1.f - b1) * gk;
                v[k] = b2 * v[k] + (1.f - b2) * gk * gk;
                p[k] -= lr * (m[k] / bc1) / (sqrtf(v[k] / bc2) + eps);
                if (do_ema) ema[k] = decay * ema[k] + (1.f - decay) * p[k];
            }
        }
    }
}

hipError_t launch_loss_grad(const float* eps_hat, const float* noise, float* d_eps, int B, int Cc, long fhw, int l2, hipStream_t st) {
    const long n = (long)B * Cc * fhw;
    const int blocks = (int)std::max<long>(1, std::min<long>((n + 255) / 256, 2048));
    hipLaunchKernelGGL(loss_grad_kernel, dim3(blocks), dim3(256), 0, st, eps_hat, noise, d_eps, B, Cc, fhw, l2, 1.0f / (float)n);
    return hipGetLastError();
}

hipError_t launch_adam_ema(float* p, const float* g, float* m, float* v, float* ema, long n, float lr, float b1, float b2, float eps,
                           long step_count, float grad_scale, int do_ema, float decay, hipStream_t st) {
    const double t = (double)step_count + 1.0;                    // optax: bias correction with count + 1
    const float bc1 = (float)(1.0 - pow((double)b1, t)), bc2 = (float)(1.0 - pow((double)b2, t));
    const int blocks = (int)std::max<long>(1, std::min<long>((n / 4 + 255) / 256, 4096));
    hipLaunchKernelGGL(adam_ema_kernel, dim3(blocks), dim3(256), 0, st, p, g, m, v, ema, n, lr, b1, b2, eps, bc1, bc2, grad_scale, do_ema, decay);
    return hipGetLastError();
}

}  // namespace vdx
