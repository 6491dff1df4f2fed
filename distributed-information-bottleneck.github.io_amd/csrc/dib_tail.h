// dib_tail.h - the END of a training / validation step in ONE launch (round 5).
//
// Until round 4 a step ended with up to eight dependent launches of a few microseconds each - the fixed-order reduce of the
// split-batch weight-gradient slabs, the reduce of the fused backward's d(W1|b1) partials, the reduce of the fused output
// head's weight-gradient partials, the KL column sums, the loss sums, the metric accumulation, Keras-Adam and its step-counter
// bump.  At the reference's default batch (train.py:30-34: B = 128) every one of them costs the 4.5-5 us a dependent kernel
// boundary costs on this part whatever it computes (profiles/r04l_default_batch_kernel_stats.csv: 7 of the 19 launches of a
// training step), and at B = 8192 - the per-GPU batch of 8-GPU strong scaling - they are 3 % of the step.
//
// One grid, five kinds of workgroups (by blockIdx range); every gradient element is produced by exactly one thread in a fixed
// summation order (deterministic, same order as the separate kernels), written to `grads` and - with DIB_TAIL_ADAM /
// DIB_TAIL_SGD - applied to the parameter in the same pass:
//   generic  float4 grid-stride over [gbeg, gend): g = sum of the nsplit partial slabs (or grads itself when there are none)
//   dw1      (feature, row k <= in_dim): d(W1|b1) = fixed-order sum of the fused backward's per-wave partials
//   head     one workgroup per element of the 1-unit output layer's (W|b): sum over the fused head's row chunks
//   kl       one workgroup per feature: KL_f local sum = column sum of the forward's per-workgroup partials -> step_out[f]
//   loss     two workgroups: task-loss sum and #correct -> step_out[F], step_out[F+1]; rows -> step_out[F+2]
// The workgroup that arrives LAST (two-level arrival counters: one atomic per workgroup spread over 32 words, so that
// arrivals do not serialise on one L2 line - 2000 workgroups on one word cost 20 us in round 2) accumulates the History
// metrics from step_out and bumps the Adam step count: every workgroup has read t by then.
#pragma once
#include "dib_common.h"

#define DIB_TAIL_LEAVES 32
#define DIB_TAIL_SYNC_WORDS (32 * (DIB_TAIL_LEAVES + 1))   // root at word 0, leaf i at word 32 (i + 1): 128 bytes apart

struct DibTailArgs {
  float* params; float* grads; float* m; float* v;
  const float* lr_dev; long long* t_dev;
  float b1, b2, eps, gscale;
  int flags;                      // include/dib_hip.h DIB_TAIL_*
  // generic segment
  long long gbeg, gend;           // element range, multiples of 4
  const float* slabs; int nsplit; long long slab_stride;   // nsplit == 0: the gradient is already in grads
  // dw1 segment (fused encoder backward)
  const float* dw1_partial; int dw1_parts, F, H1;
  const long long* w_off; const long long* b_off; const int4* featmap;
  // head segment (fused 1-unit output head)
  const float* head_partial; int head_chunks, head_K; long long head_w_off, head_b_off;
  // sums -> step_out
  const float* kl_partial; int kl_rows, kl_stride;
  const float* loss_partial; int loss_blocks; float rows;
  float* step_out;
  // metrics
  const float* beta_dev; float inv_bg; float* metrics_acc;
  unsigned* sync;
  int nb_generic, nb_dw1, nb_head, nb_kl, nb_loss;
};

struct DibTailOpt {
  DibAdamCoef c;
  float lr;
  int mode;   // 0 none, 1 Keras-Adam, 2 SGD
};

// one parameter: write the gradient, apply the optimizer (dib_adam_update: the expressions of dib_adam_kernel / dib_sgd_kernel)
__device__ __forceinline__ void dib_tail_apply1(const DibTailArgs& a, const DibTailOpt& o, long long i, float g) {
  a.grads[i] = g;
  if (o.mode == 1) {
    dib_adam_update(a.params[i], a.m[i], a.v[i], g, o.c);
  } else if (o.mode == 2) {
    a.params[i] -= o.lr * o.c.gscale * g;
  }
}

__global__ void __launch_bounds__(256)
dib_step_tail_kernel(DibTailArgs a) {
  __shared__ float red[4];
  __shared__ bool s_last;
  const int tid = threadIdx.x;
  int bid = blockIdx.x;
  DibTailOpt o;
  o.mode = (a.flags & 8) ? 1 : ((a.flags & 64) ? 2 : 0);
  o.lr = o.mode ? a.lr_dev[0] : 0.f;
  o.c = DibAdamCoef{0.f, 1.f - a.b1, 1.f - a.b2, a.eps, a.gscale};
  if (o.mode == 1) o.c = dib_adam_coef(o.lr, a.t_dev[0], a.b1, a.b2, a.eps, a.gscale);

  // block order: the few workgroups whose results the epilogue reads (KL / loss sums) come FIRST - they publish with a
  // device-scope release (an L2 write-back on this multi-XCD part) while the L2s still hold few dirty lines
  bool publishes = false;
  if (bid < a.nb_kl) {
    publishes = true;
    float s = 0.f;
    for (int i = tid; i < a.kl_rows; i += 256) s += a.kl_partial[(long long)i * a.kl_stride + bid];
    const float tot = dib_block_sum_256(s, red);
    if (tid == 0) a.step_out[bid] = tot;
  } else if ((bid -= a.nb_kl) < a.nb_loss) {
    publishes = true;
    float s = 0.f;
    for (int i = tid; i < a.loss_blocks; i += 256) s += a.loss_partial[(long long)i * 2 + bid];
    const float tot = dib_block_sum_256(s, red);
    if (tid == 0) {
      a.step_out[a.F + bid] = tot;
      if (bid == 0) a.step_out[a.F + 2] = a.rows;
    }
  } else if ((bid -= a.nb_loss) < a.nb_head) {
    const int n = a.head_K + 1;   // [chunks][K + 1]: d w[0..K) then d b
    float s = 0.f;
    for (int c = tid; c < a.head_chunks; c += 256) s += a.head_partial[(long long)c * n + bid];
    const float tot = dib_block_sum_256(s, red);
    if (tid == 0) dib_tail_apply1(a, o, bid < a.head_K ? a.head_w_off + bid : a.head_b_off, tot);
  } else if ((bid -= a.nb_head) < a.nb_dw1) {
    // grads[W1 of feature f][k][n] = sum_p partial[p][f][k][n] (k < in_dim), grads[b1][n] = sum_p partial[p][f][in_dim][n]
    const int f = bid >> 4, k = bid & 15;
    const int in_dim = a.featmap[f].y;
    if (k <= in_dim) {
      const long long pstride = (long long)a.F * 16 * a.H1;
      for (int n = tid; n < a.H1; n += 256) {
        const float* src = a.dw1_partial + ((long long)f * 16 + k) * a.H1 + n;
        float s = 0.f;
        int pz = 0;
        for (; pz + 8 <= a.dw1_parts; pz += 8) {
          float x[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) x[u] = src[(pz + u) * pstride];
#pragma unroll
          for (int u = 0; u < 8; ++u) s += x[u];
        }
        for (; pz < a.dw1_parts; ++pz) s += src[pz * pstride];
        dib_tail_apply1(a, o, k < in_dim ? a.w_off[f] + (long long)k * a.H1 + n : a.b_off[f] + n, s);
      }
    }
  } else if ((bid -= a.nb_dw1) < a.nb_generic) {
    const long long n4 = (a.gend - a.gbeg) >> 2;
    float4* G = reinterpret_cast<float4*>(a.grads + a.gbeg);
    for (long long i = bid * 256ll + tid; i < n4; i += (long long)a.nb_generic * 256) {
      float4 s;
      if (a.nsplit > 0) {
        const float4* src = reinterpret_cast<const float4*>(a.slabs + a.gbeg) + i;
        s = *src;
        for (int k = 1; k < a.nsplit; ++k) {
          const float4 x = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(src) + (long long)k * a.slab_stride);
          s.x += x.x; s.y += x.y; s.z += x.z; s.w += x.w;
        }
        G[i] = s;
      } else {
        s = G[i];
      }
      if (o.mode == 1) {
        float4 pp = reinterpret_cast<float4*>(a.params + a.gbeg)[i];
        float4 mm = reinterpret_cast<float4*>(a.m + a.gbeg)[i];
        float4 vv = reinterpret_cast<float4*>(a.v + a.gbeg)[i];
        dib_adam_update(pp.x, mm.x, vv.x, s.x, o.c);
        dib_adam_update(pp.y, mm.y, vv.y, s.y, o.c);
        dib_adam_update(pp.z, mm.z, vv.z, s.z, o.c);
        dib_adam_update(pp.w, mm.w, vv.w, s.w, o.c);
        reinterpret_cast<float4*>(a.params + a.gbeg)[i] = pp;
        reinterpret_cast<float4*>(a.m + a.gbeg)[i] = mm;
        reinterpret_cast<float4*>(a.v + a.gbeg)[i] = vv;
      } else if (o.mode == 2) {
        float4 pp = reinterpret_cast<float4*>(a.params + a.gbeg)[i];
        const float l = o.lr * o.c.gscale;
        pp.x -= l * s.x; pp.y -= l * s.y; pp.z -= l * s.z; pp.w -= l * s.w;
        reinterpret_cast<float4*>(a.params + a.gbeg)[i] = pp;
      }
    }
  }

  if (!(a.flags & (16 | 32))) return;     // nothing waits for the whole grid
  // ---- arrival: leaf = blockIdx % 32, the last of a leaf reports to the root; the last at the root owns the epilogue ----
  // Only the workgroups that wrote step_out release (every workgroup doing so - one L2 write-back each, ~900 of them at the
  // reference's default batch - made this kernel 20 us, profiles/r05c_default_batch_kernel_stats.csv); the step-count bump
  // needs no fence at all: every workgroup's read of t has returned before its arrival (its optimizer stores depend on it).
  __syncthreads();
  if (tid == 0) {
    if (publishes && (a.flags & 32)) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    const unsigned total = gridDim.x, leaf = blockIdx.x % DIB_TAIL_LEAVES;
    const unsigned leaf_n = (total - leaf + DIB_TAIL_LEAVES - 1) / DIB_TAIL_LEAVES;
    const unsigned roots = total < DIB_TAIL_LEAVES ? total : DIB_TAIL_LEAVES;
    bool last = false;
    if (__hip_atomic_fetch_add(a.sync + 32 * (leaf + 1), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == leaf_n - 1) {
      __hip_atomic_store(a.sync + 32 * (leaf + 1), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // self-cleaning
      if (__hip_atomic_fetch_add(a.sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == roots - 1) {
        __hip_atomic_store(a.sync, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = true;
      }
    }
    s_last = last;
  }
  __syncthreads();
  if (!s_last) return;
  if (a.flags & 32) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  if (a.flags & 32) {
    // step_out: [0..F) KL local sums, [F] task-loss local sum, [F+1] #correct, [F+2] rows (dib_metrics_accumulate_kernel)
    for (int i = tid; i < a.F + 3; i += 256) {
      if (i < a.F) a.metrics_acc[i] += __hip_atomic_load(a.step_out + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) * a.inv_bg;
      else if (i == a.F) {
        float s = 0.f;
        for (int f = 0; f < a.F; ++f) s += __hip_atomic_load(a.step_out + f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        a.metrics_acc[a.F] += __hip_atomic_load(a.step_out + a.F, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + a.beta_dev[0] * s;
      } else a.metrics_acc[i] += __hip_atomic_load(a.step_out + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if ((a.flags & 16) && tid == 0) a.t_dev[0] += 1;
}
