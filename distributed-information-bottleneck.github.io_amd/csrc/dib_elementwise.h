// dib_elementwise.h - the HBM-bound kernels of the Distributed-IB path (gfx950).
//   positional encoding (+ batch gather), fused Gaussian reparameterisation + KL-to-unit-prior,
//   its backward, task losses, metric accumulation, split-batch gradient reduce, Keras-Adam.
#pragma once
#include "dib_common.h"

// ---------------------------------------------------------------------------------------------
// Positional encoding, reference models.py:22-23:  concat([x] + [sin(f*x) for f in freqs], -1)
// (blockwise layout) fused with tf.split (models.py:101) and the shuffled-batch gather.
// colmap[c] = {feature, local column, d_f, sum of encoder-input widths of the features before it}.
// P is feature-major and ragged: P_f = P + poff*batch is a dense [batch, n_blocks*d_f] matrix and
// P_f[b, j*d_f + c_local] = (j==0 ? x : sin(2^j * x)),  x = X[row(b), c]
// ---------------------------------------------------------------------------------------------
template <int ROWS>   // batch rows per workgroup tile: 64, or 16 for mid-size batches (more workgroups: at B = 8192 the 64-row
                      // tiling is 128 workgroups of sinf-bound work on 256 CUs)
__global__ void __launch_bounds__(256)
dib_posenc_kernel(const float* __restrict__ X, long long ldx, const int* __restrict__ row_idx, long long row0,
                  int batch, const int4* __restrict__ colmap, int ncols, int n_blocks /*1 + n sinusoids*/,
                  float* __restrict__ P) {
  // One block = ROWS rows x 64 input columns.  The tile is read row-wise (coalesced along the sample-major X rows),
  // transposed through LDS, and written with lanes <-> consecutive rows of ONE feature, so the feature-major P rows
  // (width*4 bytes apart) are filled by neighbouring lanes instead of 4-byte stores scattered over 64 features.
  __shared__ float T[ROWS][65];
  const int c0 = blockIdx.x * 64, b0 = blockIdx.y * ROWS;
#pragma unroll
  for (int i = 0; i < ROWS / 4; ++i) {
    const int idx = threadIdx.x + 256 * i, r = idx >> 6, c = idx & 63;
    float v = 0.f;
    if (b0 + r < batch && c0 + c < ncols) {
      const long long row = row_idx ? (long long)row_idx[b0 + r] : row0 + b0 + r;
      v = X[row * ldx + c0 + c];
    }
    T[r][c] = v;
  }
  __syncthreads();
  const int r = threadIdx.x & (ROWS - 1), b = b0 + r;
  if (b >= batch) return;
  for (int c = threadIdx.x / ROWS; c < 64 && c0 + c < ncols; c += 256 / ROWS) {
    const int4 cm = colmap[c0 + c];
    const float x = T[r][c];
    float* dst = P + (long long)cm.w * batch + (long long)b * (n_blocks * cm.z) + cm.y;
    dst[0] = x;
    float fr = 2.0f;
    for (int j = 1; j < n_blocks; ++j) {
      dst[(long long)j * cm.z] = sinf(fr * x);  // accurate sinf (range-reduced), |fr*x| can reach ~100
      fr *= 2.0f;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Fused reparameterisation + KL, reference models.py:106-112:
//   mu, logvar = split(enc_out_f, 2) ; u = mu + exp(logvar/2)*eps ; KL_f = mean_b sum_e 0.5(mu^2+e^lv-lv-1)
// One block = (row tile, feature); thread = (row, 4 consecutive dims) -> one Philox call.
// KL partial sums are written per block (second stage sums them in a fixed order: deterministic).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
dib_reparam_kl_fwd_kernel(const float* __restrict__ enc_out, float* __restrict__ U, float* __restrict__ kl_partial,
                          const int* __restrict__ row_idx, long long row0, int batch, int F, int E,
                          unsigned long long seed, unsigned step, int deterministic, const unsigned* step_dev,
                          float lv_off = 0.f /* set transformer: logvar += -3 before sampling / KL */) {
  __shared__ float red[4];
  if (step_dev) step = step_dev[0];
  const int E4 = (E + 3) >> 2;
  const int rows_per_block = 256 / E4;
  const int f = blockIdx.y;
  const int r = threadIdx.x / E4, q = threadIdx.x - r * E4;
  const int b = blockIdx.x * rows_per_block + r;
  float klp = 0.f;
  if (r < rows_per_block && b < batch) {
    const long long grow = row_idx ? (long long)row_idx[b] : row0 + b;
    const float* mu_p = enc_out + ((long long)f * batch + b) * (2ll * E) + 4 * q;  // enc_out is [F][B][2E]
    const float* lv_p = mu_p + E;
    float* u_p = U + (long long)b * ((long long)F * E) + (long long)f * E + 4 * q;
    float eps[4] = {0.f, 0.f, 0.f, 0.f};
    if (!deterministic) dib_eps4(seed, step, (uint32_t)grow, (uint32_t)f, (uint32_t)q, eps);
    if ((E & 3) == 0) {
      const float4 mu = *reinterpret_cast<const float4*>(mu_p);
      float4 lv = *reinterpret_cast<const float4*>(lv_p);
      lv.x += lv_off; lv.y += lv_off; lv.z += lv_off; lv.w += lv_off;
      float4 u;
      u.x = mu.x + expf(0.5f * lv.x) * eps[0];
      u.y = mu.y + expf(0.5f * lv.y) * eps[1];
      u.z = mu.z + expf(0.5f * lv.z) * eps[2];
      u.w = mu.w + expf(0.5f * lv.w) * eps[3];
      *reinterpret_cast<float4*>(u_p) = u;
      klp = 0.5f * ((mu.x * mu.x + expf(lv.x) - lv.x - 1.f) + (mu.y * mu.y + expf(lv.y) - lv.y - 1.f) +
                    (mu.z * mu.z + expf(lv.z) - lv.z - 1.f) + (mu.w * mu.w + expf(lv.w) - lv.w - 1.f));
    } else {
      for (int j = 0; j < 4; ++j) {
        if (4 * q + j < E) {
          const float mu = mu_p[j], lv = lv_p[j] + lv_off;
          u_p[j] = mu + expf(0.5f * lv) * eps[j];
          klp += 0.5f * (mu * mu + expf(lv) - lv - 1.f);
        }
      }
    }
  }
  const float tot = dib_block_sum_256(klp, red);
  if (threadIdx.x == 0) kl_partial[(long long)blockIdx.x * F + f] = tot;
}

// second stage: out[f] = sum_blocks partial[blk][f]   (one block per feature, fixed order)
__global__ void __launch_bounds__(256)
dib_colsum_partials_kernel(const float* __restrict__ partial, int nblocks, int stride, float* __restrict__ out) {
  __shared__ float red[4];
  const int f = blockIdx.x;
  float s = 0.f;
  for (int i = threadIdx.x; i < nblocks; i += 256) s += partial[(long long)i * stride + f];
  const float tot = dib_block_sum_256(s, red);
  if (threadIdx.x == 0) out[f] = tot;
}

// Backward of reparam + KL (what tape.gradient derives from models.py:108,111-112,118):
//   dmu = g_u + beta*mu/Bg ;  dlogvar = g_u*(eps*sigma)*0.5 + beta*0.5*(exp(lv)-1)/Bg
// The noise term is recovered from the forward's own sample:  eps*sigma = u - mu  (u = fl(mu + fl(sigma*eps)), so the
// difference carries at most one ulp of u; it is multiplied by 0.5*g_u).  Nothing is regenerated, and the gradient is by
// construction the gradient of the forward that actually ran - library noise, caller-injected samples or the
// deterministic forward (u = mu: the noise term vanishes).  U is sample-major [B][F*E] like GU.
__global__ void __launch_bounds__(256)
dib_reparam_kl_bwd_kernel(const float* __restrict__ enc_out, const float* __restrict__ GU, const float* __restrict__ U,
                          float* __restrict__ dout, const float* __restrict__ beta_dev, float inv_bg, int batch, int F, int E,
                          float lv_off = 0.f) {
  const int E4 = (E + 3) >> 2;
  const int rows_per_block = 256 / E4;
  const int f = blockIdx.y;
  const int r = threadIdx.x / E4, q = threadIdx.x - r * E4;
  const int b = blockIdx.x * rows_per_block + r;
  if (r >= rows_per_block || b >= batch) return;
  const float kb = beta_dev[0] * inv_bg;
  const long long o = ((long long)f * batch + b) * (2ll * E) + 4 * q;  // enc_out / dout are [F][B][2E]
  const long long so = (long long)b * ((long long)F * E) + (long long)f * E + 4 * q;
  const float* gu_p = GU + so;
  const float* u_p = U + so;
  if ((E & 3) == 0) {
    const float4 mu = *reinterpret_cast<const float4*>(enc_out + o);
    float4 lv = *reinterpret_cast<const float4*>(enc_out + o + E);
    lv.x += lv_off; lv.y += lv_off; lv.z += lv_off; lv.w += lv_off;
    const float4 gu = *reinterpret_cast<const float4*>(gu_p);
    const float4 u = *reinterpret_cast<const float4*>(u_p);
    float4 dm, dl;
    dm.x = gu.x + kb * mu.x; dm.y = gu.y + kb * mu.y; dm.z = gu.z + kb * mu.z; dm.w = gu.w + kb * mu.w;
    dl.x = gu.x * (u.x - mu.x) * 0.5f + kb * 0.5f * (expf(lv.x) - 1.f);
    dl.y = gu.y * (u.y - mu.y) * 0.5f + kb * 0.5f * (expf(lv.y) - 1.f);
    dl.z = gu.z * (u.z - mu.z) * 0.5f + kb * 0.5f * (expf(lv.z) - 1.f);
    dl.w = gu.w * (u.w - mu.w) * 0.5f + kb * 0.5f * (expf(lv.w) - 1.f);
    *reinterpret_cast<float4*>(dout + o) = dm;
    *reinterpret_cast<float4*>(dout + o + E) = dl;
  } else {
    for (int j = 0; j < 4; ++j) {
      if (4 * q + j < E) {
        const float mu = enc_out[o + j], lv = enc_out[o + E + j] + lv_off, gu = gu_p[j];
        dout[o + j] = gu + kb * mu;
        dout[o + E + j] = gu * (u_p[j] - mu) * 0.5f + kb * 0.5f * (expf(lv) - 1.f);
      }
    }
  }
}

__global__ void __launch_bounds__(256)
dib_eps_fill_kernel(float* __restrict__ eps_out, const int* __restrict__ row_idx, long long row0, int batch, int F,
                    int E, unsigned long long seed, unsigned step) {
  const int E4 = (E + 3) >> 2;
  const long long total = (long long)batch * F * E4;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int q = (int)(i % E4);
    const int f = (int)((i / E4) % F);
    const int b = (int)(i / ((long long)E4 * F));
    const long long grow = row_idx ? (long long)row_idx[b] : row0 + b;
    float e[4];
    dib_eps4(seed, step, (uint32_t)grow, (uint32_t)f, (uint32_t)q, e);
    for (int j = 0; j < 4; ++j)
      if (4 * q + j < E) eps_out[((long long)b * F + f) * E + 4 * q + j] = e[j];
  }
}

// ---------------------------------------------------------------------------------------------
// Task loss + its gradient wrt the model output (Keras semantics).
//   kind 0: BinaryCrossentropy(from_logits=True) (reference data.py:65): max(z,0) - z*y + log1p(exp(-|z|))
//   kind 1: BinaryCrossentropy on probabilities (clip 1e-7)
//   kind 2: SparseCategoricalCrossentropy(from_logits=True) (reference data.py:343)
//   kind 3: 'mse'
// One thread per batch row.  Writes g_pred (already multiplied by act'(pred) of the output
// activation), per-block partial {loss sum, #correct}.  inv_bg = 1/B_global.
// 'accuracy' follows Keras: 1-unit output -> binary accuracy thresholding the raw output at 0.5,
// sparse labels -> argmax match.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
dib_loss_kernel(int kind, const float* __restrict__ pred, int out_dim, const float* __restrict__ Y, long long ldy,
                const int* __restrict__ row_idx, long long row0, int batch, float inv_bg, int out_act,
                float* __restrict__ g_pred, float* __restrict__ partial /*[gridDim.x][2]*/) {
  __shared__ float red[4];
  const int b = blockIdx.x * 256 + threadIdx.x;
  float lsum = 0.f, correct = 0.f;
  if (b < batch) {
    const long long row = row_idx ? (long long)row_idx[b] : row0 + b;
    const float* p = pred + (long long)b * out_dim;
    float* g = g_pred + (long long)b * out_dim;
    const float* y = Y + row * ldy;
    if (kind == 0 || kind == 1 || kind == 3) {
      const float sc = inv_bg / (float)out_dim;
      float nright = 0.f;
      for (int o = 0; o < out_dim; ++o) {
        const float z = p[o], yy = y[o];
        float l, gg;
        if (kind == 0) {
          l = fmaxf(z, 0.f) - z * yy + log1pf(expf(-fabsf(z)));
          gg = 1.0f / (1.0f + expf(-z)) - yy;
        } else if (kind == 1) {
          const float pc = fminf(fmaxf(z, 1e-7f), 1.0f - 1e-7f);
          l = -(yy * logf(pc) + (1.f - yy) * logf(1.f - pc));
          gg = (z < 1e-7f || z > 1.0f - 1e-7f) ? 0.f : (-(yy / pc) + (1.f - yy) / (1.f - pc));
        } else {
          const float d = z - yy;
          l = d * d;
          gg = 2.f * d;
        }
        lsum += l;
        g[o] = gg * sc * dib_act_grad(out_act, z);
        nright += ((z > 0.5f ? 1.f : 0.f) == yy) ? 1.f : 0.f;
      }
      lsum /= (float)out_dim;
      correct = nright / (float)out_dim;
    } else {  // sparse categorical cross-entropy from logits
      const int lab = (int)y[0];
      float m = -INFINITY;
      int am = 0;
      for (int o = 0; o < out_dim; ++o)
        if (p[o] > m) { m = p[o]; am = o; }
      float se = 0.f;
      for (int o = 0; o < out_dim; ++o) se += expf(p[o] - m);
      const float lse = m + logf(se);
      lsum = lse - p[lab];
      for (int o = 0; o < out_dim; ++o) g[o] = (expf(p[o] - lse) - (o == lab ? 1.f : 0.f)) * inv_bg;
      correct = (am == lab) ? 1.f : 0.f;
    }
  }
  const float tl = dib_block_sum_256(lsum, red);
  const float tc = dib_block_sum_256(correct, red);
  if (threadIdx.x == 0) {
    partial[2 * blockIdx.x] = tl;
    partial[2 * blockIdx.x + 1] = tc;
  }
}

// step_out layout: [0..F) KL local sums, [F] task-loss local sum, [F+1] #correct, [F+2] rows
// metrics_acc[f]   += KL_f_sum * inv_bg                       (History 'KL{f}', reference models.py:115)
// metrics_acc[F]   += task_sum + beta * sum_f KL_f_sum        ('loss' incl. beta*KL, models.py:118)
// metrics_acc[F+1] += #correct ; metrics_acc[F+2] += rows
__global__ void dib_metrics_accumulate_kernel(const float* __restrict__ step_out, int F, const float* beta_dev,
                                              float inv_bg, float* __restrict__ acc) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < F) acc[i] += step_out[i] * inv_bg;
  if (i == F) {
    float s = 0.f;
    for (int f = 0; f < F; ++f) s += step_out[f];
    acc[F] += step_out[F] + beta_dev[0] * s;
  }
  if (i == F + 1) acc[F + 1] += step_out[F + 1];
  if (i == F + 2) acc[F + 2] += step_out[F + 2];
}


// out[i] = (acc ? acc[i] : 0) + sum_s partial[s*stride + i], i < n   (fixed order => deterministic; acc may be out itself)
__global__ void __launch_bounds__(256)
dib_reduce_splits_kernel(const float* __restrict__ partial, long long n, int nsplit, long long stride,
                         float* out, const float* acc = nullptr) {
  const long long n4 = n >> 2;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4;
       i += (long long)gridDim.x * blockDim.x) {
    float4 s = reinterpret_cast<const float4*>(partial)[i];
    for (int k = 1; k < nsplit; ++k) {
      const float4 v = reinterpret_cast<const float4*>(partial + (long long)k * stride)[i];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    if (acc) {
      const float4 a = reinterpret_cast<const float4*>(acc)[i];
      s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
    }
    reinterpret_cast<float4*>(out)[i] = s;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const long long i = (n4 << 2) + threadIdx.x;
    float s = 0.f;
    for (int k = 0; k < nsplit; ++k) s += partial[(long long)k * stride + i];
    out[i] = acc ? s + acc[i] : s;
  }
}

// ---------------------------------------------------------------------------------------------
// Keras Adam (reference train.py:128-129 tf.keras.optimizers.get('adam')):
//   m += (1-b1)(g-m); v += (1-b2)(g^2-v); theta -= lr*sqrt(1-b2^t)/(1-b1^t) * m/(sqrt(v)+eps)
// t = *t_dev + 1 (device counter, bumped by dib_bump_counter_kernel afterwards).  (Bumping it inside this kernel through a
// last-workgroup-done arrival counter was tried in round 2 and reverted: 2173 workgroups x one atomicAdd on one word
// serialise at ~11 ns each - the kernel went from 14 to 35 us to save a 4.5 us launch.)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
dib_adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                long long n, const float* __restrict__ lr_dev, const long long* __restrict__ t_dev, float b1,
                float b2, float eps, float gscale) {
  const DibAdamCoef c = dib_adam_coef(lr_dev[0], t_dev[0], b1, b2, eps, gscale);
  const long long n4 = n >> 2;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4;
       i += (long long)gridDim.x * blockDim.x) {
    float4 pp = reinterpret_cast<float4*>(p)[i];
    const float4 gg = reinterpret_cast<const float4*>(g)[i];
    float4 mm = reinterpret_cast<float4*>(m)[i];
    float4 vv = reinterpret_cast<float4*>(v)[i];
    dib_adam_update(pp.x, mm.x, vv.x, gg.x, c);
    dib_adam_update(pp.y, mm.y, vv.y, gg.y, c);
    dib_adam_update(pp.z, mm.z, vv.z, gg.z, c);
    dib_adam_update(pp.w, mm.w, vv.w, gg.w, c);
    reinterpret_cast<float4*>(p)[i] = pp;
    reinterpret_cast<float4*>(m)[i] = mm;
    reinterpret_cast<float4*>(v)[i] = vv;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const long long i = (n4 << 2) + threadIdx.x;
    dib_adam_update(p[i], m[i], v[i], g[i], c);
  }
}

__global__ void dib_bump_counter_kernel(long long* t) { t[0] += 1; }

// loss second stage: step_out[F] = task-loss sum, step_out[F+1] = #correct (fixed-order sums of the per-block partials),
// step_out[F+2] = rows.  grid = 2 workgroups.
__global__ void __launch_bounds__(256)
dib_loss_finalize_kernel(const float* __restrict__ partial, int nblocks, float rows, float* __restrict__ out) {
  __shared__ float red[4];
  const int f = blockIdx.x;
  float s = 0.f;
  for (int i = threadIdx.x; i < nblocks; i += 256) s += partial[(long long)i * 2 + f];
  const float tot = dib_block_sum_256(s, red);
  if (threadIdx.x == 0) {
    out[f] = tot;
    if (f == 0) out[2] = rows;
  }
}

__global__ void __launch_bounds__(256)
dib_sgd_kernel(float* __restrict__ p, const float* __restrict__ g, long long n, const float* __restrict__ lr_dev,
               float gscale) {
  const float lr = lr_dev[0] * gscale;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x)
    p[i] -= lr * g[i];
}

// Bhattacharyya distance between diagonal Gaussians (reference utils.py:177-212; closed form instead of the
// reference's dense [N,M,d,d] diagonal matrices): D = 1/8 sum dmu^2/sbar + 1/2 sum ln(sbar) - 1/4 (sum lv1 + sum lv2)
__global__ void __launch_bounds__(256)
dib_bhattacharyya_kernel(const float* __restrict__ mu1, const float* __restrict__ lv1, int n,
                         const float* __restrict__ mu2, const float* __restrict__ lv2, int m, int dim,
                         float* __restrict__ out) {
  const long long total = (long long)n * m;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int i = (int)(idx / m), j = (int)(idx - (long long)i * m);
    float t1 = 0.f, t2 = 0.f;
    for (int e = 0; e < dim; ++e) {
      const float a = mu1[(long long)i * dim + e], b = mu2[(long long)j * dim + e];
      const float la = lv1[(long long)i * dim + e], lb = lv2[(long long)j * dim + e];
      const float sb = 0.5f * (expf(la) + expf(lb));
      const float d = a - b;
      t1 += d * d / sb;
      t2 += logf(sb) - 0.5f * (la + lb);
    }
    out[idx] = 0.125f * t1 + 0.5f * t2;
  }
}

// ---------------------------------------------------------------------------------------------
// Mutual-information sandwich bounds (reference utils.py:10-73, Poole et al. 2019): for a batch of N
// encoded points with p(u|x_j) = N(mu_j, diag(exp(logvar_j))) and one sample u_i ~ p(u|x_i),
//   log p_ij = -1/2 sum_e ((u_ie - mu_je)/sigma_je)^2 - 1/2 sum_e logvar_je - E/2 ln(2 pi)
//   InfoNCE lower_i = log p_ii - log( 1/N sum_j    p_ij )
//   leave-one-out upper_i = log p_ii - log( 1/N sum_{j!=i} p_ij )      (the reference divides by N, not N-1)
// Everything in float64 like the reference (utils.py:39-40), but with a log-sum-exp so that well separated
// Gaussians do not underflow to log(0) as the reference's exp-then-log does.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
dib_mi_prep_kernel(const float* __restrict__ enc_out /*[N][2E]*/, int n, int E, unsigned long long seed, unsigned step,
                   unsigned feature, double* __restrict__ inv_sigma /*[N][E]*/, double* __restrict__ u /*[N][E]*/,
                   double* __restrict__ cj /*[N]*/, double* __restrict__ mu_t /*[E][N]*/, double* __restrict__ is_t /*[E][N]*/,
                   float lv_off = 0.f /* set transformer: logvar - 3 */) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const float* mu = enc_out + (long long)j * 2 * E;
  const float* lv = mu + E;
  double slv = 0.0;
  for (int q = 0; q < (E + 3) / 4; ++q) {
    float eps[4];
    dib_eps4(seed, step, (uint32_t)j, feature, (uint32_t)q, eps);
    for (int t = 0; t < 4 && 4 * q + t < E; ++t) {
      const int e = 4 * q + t;
      const double l = (double)lv[e] + (double)lv_off;
      const double sd = exp(0.5 * l);
      inv_sigma[(long long)j * E + e] = 1.0 / sd;
      u[(long long)j * E + e] = (double)mu[e] + sd * (double)eps[t];
      // dimension-major copies for the row kernels: thread j of a row's workgroup reads [e][j] - consecutive threads,
      // consecutive addresses (the point-major arrays made every load of the N^2 E inner loop touch 64 cache lines per
      // wave: 218 us per 1024 x 1024 x 32 evaluation; round 3)
      mu_t[(long long)e * n + j] = (double)mu[e];
      is_t[(long long)e * n + j] = 1.0 / sd;
      slv += l;
    }
  }
  cj[j] = -0.5 * slv - 0.5 * (double)E * 1.8378770664093454835606594728112;  // ln(2 pi)
}

__device__ __forceinline__ void dib_lse_add(double& mx, double& sm, double v) {
  if (v > mx) { sm = sm * exp(mx - v) + 1.0; mx = v; }
  else sm += exp(v - mx);
}

__global__ void __launch_bounds__(256)
dib_mi_rows_kernel(const float* __restrict__ enc_out, int n, int E, const double* __restrict__ inv_sigma,
                   const double* __restrict__ u, const double* __restrict__ cj, const double* __restrict__ mu_t,
                   const double* __restrict__ is_t, double* __restrict__ lower_rows, double* __restrict__ upper_rows) {
  __shared__ double smx[256], ssm[256];
  const int i = blockIdx.x;
  const double* ui = u + (long long)i * E;
  double mx = -1.0e300, sm = 0.0;  // log-sum-exp over j != i
  for (int j = threadIdx.x; j < n; j += 256) {
    if (j == i) continue;
    double q = 0.0;
    for (int e = 0; e < E; ++e) {
      const double d = (ui[e] - mu_t[(long long)e * n + j]) * is_t[(long long)e * n + j];
      q = fma(d, d, q);
    }
    dib_lse_add(mx, sm, cj[j] - 0.5 * q);
  }
  smx[threadIdx.x] = mx;
  ssm[threadIdx.x] = sm;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      const double m1 = smx[threadIdx.x], m2 = smx[threadIdx.x + s];
      const double s1 = ssm[threadIdx.x], s2 = ssm[threadIdx.x + s];
      const double m = m1 > m2 ? m1 : m2;
      smx[threadIdx.x] = m;
      ssm[threadIdx.x] = s1 * exp(m1 - m) + s2 * exp(m2 - m);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float* mu = enc_out + (long long)i * 2 * E;
    const double* is = inv_sigma + (long long)i * E;
    double q = 0.0;
    for (int e = 0; e < E; ++e) {
      const double d = (ui[e] - (double)mu[e]) * is[e];
      q = fma(d, d, q);
    }
    const double lii = cj[i] - 0.5 * q;
    const double lse_off = (ssm[0] > 0.0) ? smx[0] + log(ssm[0]) : -INFINITY;
    const double mall = lii > lse_off ? lii : lse_off;
    const double lse_all = mall + log(exp(lii - mall) + exp(lse_off - mall));
    const double logn = log((double)n);
    lower_rows[i] = lii - (lse_all - logn);
    upper_rows[i] = lii - (lse_off - logn);
  }
}


// Per-particle information map of the set-transformer notebook (probe-grid MI bounds, cell 8 "Now use probe points along
// with a bunch of real points ..."): M probe Gaussians with one sample u_i each, N data Gaussians,
//   lii = log p(u_i | probe_i),  lij = log p(u_i | data_j)
//   lower_i = lii - ( LSE(lii, li1 .. liN) - log(N + 1) )       infonce_per (N + 1 terms in the mean)
//   upper_i = lii - ( LSE(li1 .. liN)      - log N )            loo_per
// float64 with a log-sum-exp (the notebook's exp-then-log underflows for separated Gaussians).  One workgroup per probe.
__global__ void __launch_bounds__(256)
dib_mi_probe_rows_kernel(const float* __restrict__ enc_probe, const double* __restrict__ u_probe,
                         const double* __restrict__ is_probe, const double* __restrict__ c_probe,
                         const double* __restrict__ mu_t_data /*[E][n_data]*/, const double* __restrict__ is_t_data,
                         const double* __restrict__ c_data, int n_data, int E, double* __restrict__ lower_rows,
                         double* __restrict__ upper_rows) {
  __shared__ double smx[256], ssm[256];
  const int i = blockIdx.x;
  const double* ui = u_probe + (long long)i * E;
  double mx = -1.0e300, sm = 0.0;
  for (int j = threadIdx.x; j < n_data; j += 256) {
    double q = 0.0;
    for (int e = 0; e < E; ++e) {
      const double d = (ui[e] - mu_t_data[(long long)e * n_data + j]) * is_t_data[(long long)e * n_data + j];
      q = fma(d, d, q);
    }
    dib_lse_add(mx, sm, c_data[j] - 0.5 * q);
  }
  smx[threadIdx.x] = mx;
  ssm[threadIdx.x] = sm;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      const double m1 = smx[threadIdx.x], m2 = smx[threadIdx.x + s];
      const double s1 = ssm[threadIdx.x], s2 = ssm[threadIdx.x + s];
      const double m = m1 > m2 ? m1 : m2;
      smx[threadIdx.x] = m;
      ssm[threadIdx.x] = s1 * exp(m1 - m) + s2 * exp(m2 - m);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float* mu = enc_probe + (long long)i * 2 * E;
    const double* is = is_probe + (long long)i * E;
    double q = 0.0;
    for (int e = 0; e < E; ++e) {
      const double d = (ui[e] - (double)mu[e]) * is[e];
      q = fma(d, d, q);
    }
    const double lii = c_probe[i] - 0.5 * q;
    const double lse_data = (ssm[0] > 0.0) ? smx[0] + log(ssm[0]) : -INFINITY;
    const double mall = lii > lse_data ? lii : lse_data;
    const double lse_all = mall + log(exp(lii - mall) + exp(lse_data - mall));
    lower_rows[i] = lii - (lse_all - log((double)n_data + 1.0));
    upper_rows[i] = lii - (lse_data - log((double)n_data));
  }
}

// ---------------------------------------------------------------------------------------------
// InfoNCE path (reference train.py:201-220 eval_batch_infonce; utils.py:75-175 get_scaled_similarity):
//   S = similarity(emb_x, emb_y) / T  [B,B] ;  loss = mean_i CE(i, S[i,:]) + mean_j CE(j, S[:,j])  (NOT halved)
// similarity ids: 0 l2sq, 1 l2, 2 l1, 3 linf, 4 cosine.  B is the batch (128 by default, 2048 in the chaos notebook), D the
// shared embedding width (64 by default).  Deterministic VALU kernels, O(B^2 D) each:
//   norms   |x_i|^2, |y_j|^2 once per row (l2sq / l2 / cosine)
//   sim     32 x 32 pairs per workgroup, both 32-row embedding tiles in LDS (odd pitch: conflict-free), one dot product /
//           L1 / Linf reduction per pair; Linf also records its arg-max coordinate; S and its transpose are both written
//   lse     row log-sum-exp of S and of ST (= column log-sum-exp);   loss
//   coef    per-pair gradient coefficient(s) from dL/dS_ij = (softmax_row + softmax_col - 2 delta) / (B T), both orientations
//   grad    g_x[i][e] = sum_j coef_ij (x_ie - y_je)-type sums (and g_y over the transposed copies): thread = (coordinate e,
//           partner group), coefficients of 256 partners at a time through LDS.
// (Round 3: the first version recomputed the norms and the dot product inside the gradient loop - O(B^2 D^2), 9.0 ms per
// call at B = 2048, D = 64 against 0.35 ms for the whole encoder step; tools/infonce_bench.py, profiles/r03i_*, r03j_*.)
// ---------------------------------------------------------------------------------------------
// grid (ceil(B/32) column tiles, ceil(B/32) row tiles), 256 threads, dynamic LDS 2 * 32 * (D + 1) floats.
// Writes S AND its transpose ST (and the arg-max table and its transpose): every later pass over columns - column
// log-sum-exp, g_y - then reads rows of the transposed copy, coalesced (the first version walked S with stride B).
__global__ void __launch_bounds__(256)
dib_infonce_sim_kernel(const float* __restrict__ X, const float* __restrict__ Y, int B, int D, int kind, float inv_t,
                       const float* __restrict__ norms, float* __restrict__ S, float* __restrict__ ST,
                       int* __restrict__ amax, int* __restrict__ amaxT) {
  extern __shared__ float sm[];
  __shared__ float tile[32][33];
  __shared__ int atile[32][33];
  const int P = D + 1;
  float* Xs = sm;
  float* Ys = sm + 32 * P;
  const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
  for (int idx = threadIdx.x; idx < 32 * D; idx += 256) {
    const int r = idx / D, e = idx - r * D;
    Xs[r * P + e] = (i0 + r < B) ? X[(long long)(i0 + r) * D + e] : 0.f;
    Ys[r * P + e] = (j0 + r < B) ? Y[(long long)(j0 + r) * D + e] : 0.f;
  }
  __syncthreads();
  const int tj = threadIdx.x & 31, ti = threadIdx.x >> 5;
  const int j = j0 + tj;
  const float* b = Ys + tj * P;
  for (int k = 0; k < 4; ++k) {
    const int il = ti + 8 * k, i = i0 + il;
    const float* a = Xs + il * P;
    float s = 0.f;
    int am = 0;
    if (kind == 0 || kind == 1 || kind == 4) {
      float ab = 0.f;
      for (int e = 0; e < D; ++e) ab += a[e] * b[e];
      const float na = norms[min(i, B - 1)], nb = norms[B + min(j, B - 1)];
      if (kind == 4) {
        s = ab / (sqrtf(na) * sqrtf(nb));
      } else {  // utils.py:85-90: max(|a|^2 + |b|^2 - 2 a.b, 0)
        const float d2 = fmaxf(na + nb - 2.0f * ab, 0.f);
        s = (kind == 0) ? -d2 : -sqrtf(d2 + 1e-9f);
      }
    } else if (kind == 2) {
      for (int e = 0; e < D; ++e) s -= fabsf(a[e] - b[e]);
    } else {
      float mx = -1.f;
      for (int e = 0; e < D; ++e) { const float v = fabsf(a[e] - b[e]); if (v > mx) { mx = v; am = e; } }  // first maximum
      s = -mx;
    }
    s *= inv_t;
    tile[il][tj] = s;
    atile[il][tj] = am;
    if (i < B && j < B) {
      S[(long long)i * B + j] = s;
      if (kind == 3) amax[(long long)i * B + j] = am;
    }
  }
  __syncthreads();
  for (int k = 0; k < 4; ++k) {   // transposed tile: row = column index j0 + .., consecutive lanes = consecutive i
    const int jl = ti + 8 * k, jj = j0 + jl, ii = i0 + tj;
    if (jj < B && ii < B) {
      ST[(long long)jj * B + ii] = tile[tj][jl];
      if (kind == 3) amaxT[(long long)jj * B + ii] = atile[tj][jl];
    }
  }
}

// lse[0][i] = LSE_j S[i][j] (rows of S), lse[1][j] = LSE_i S[i][j] (rows of ST); one block per row
__global__ void __launch_bounds__(256)
dib_infonce_lse_kernel(const float* __restrict__ S, const float* __restrict__ ST, int B, float* __restrict__ lse) {
  __shared__ float red[4];
  const int which = blockIdx.y, idx = blockIdx.x;
  const float* row = (which == 0 ? S : ST) + (long long)idx * B;
  float mx = -INFINITY;
  for (int t = threadIdx.x; t < B; t += 256) mx = fmaxf(mx, row[t]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sm = 0.f;
  for (int t = threadIdx.x; t < B; t += 256) sm += expf(row[t] - mx);
  const float tot = dib_block_sum_256(sm, red);
  if (threadIdx.x == 0) lse[(long long)which * B + idx] = mx + logf(tot);
}

// loss = (1/B) sum_i (lse_r[i] - S_ii) + (1/B) sum_j (lse_c[j] - S_jj)
__global__ void __launch_bounds__(256)
dib_infonce_loss_kernel(const float* __restrict__ S, const float* __restrict__ lse, int B, float* __restrict__ loss_out) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < B; i += 256) s += lse[i] + lse[B + i] - 2.0f * S[(long long)i * B + i];
  const float tot = dib_block_sum_256(s, red);
  if (threadIdx.x == 0) loss_out[0] = tot / (float)B;
}

// Per-pair gradient coefficient, for BOTH orientations (blockIdx.y = 0: C[i][j] from S; 1: CT[j][i] from ST - the same
// numbers, each written along its own rows).  With w_ij = dL/d(unscaled similarity)_ij = (softmax_row_i(S)_ij +
// softmax_col_j(S)_ij - 2 delta_ij) / (B T), the derivative of the similarity w.r.t. coordinate e of the row's own embedding
// a (partner b) factors as
//   l2sq  -2 (a_e - b_e)            if d2 > 0  (S = -d2/T < 0)                 c = -2 w
//   l2    -(a_e - b_e) / r          r = sqrt(d2 + 1e-9) = -S T, if r^2 > 1e-9   c = -w / r
//   l1    -sign(a_e - b_e)                                                      c = -w
//   linf  -sign(a_e - b_e) [e == argmax]                                        c = -w
//   cos   (b_e/|b| - sim a_e/|a|) / |a|    c = w / (|a| |b|),  c2 = w sim / |a|^2  (second output plane)
// so the gradient kernel spends 2-4 instructions per (pair, coordinate) instead of re-deriving this per coordinate.
__global__ void __launch_bounds__(256)
dib_infonce_coef_kernel(const float* __restrict__ S, const float* __restrict__ ST, const float* __restrict__ lse,
                        const float* __restrict__ norms, int B, int kind, float inv_t, float temperature,
                        float* __restrict__ C, float* __restrict__ CT, float* __restrict__ C2, float* __restrict__ C2T) {
  const long long total = (long long)B * B;
  const int tr = blockIdx.y;                    // 0: rows = x index i; 1: rows = y index j
  const float* Sm = tr == 0 ? S : ST;
  float* Cm = tr == 0 ? C : CT;
  float* C2m = tr == 0 ? C2 : C2T;
  const float sc = inv_t / (float)B;
  for (long long idx = blockIdx.x * 256ll + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int r = (int)(idx / B), c = (int)(idx - (long long)r * B);
    const int i = tr == 0 ? r : c, j = tr == 0 ? c : r;
    const float sij = Sm[idx];
    const float w = (expf(sij - lse[i]) + expf(sij - lse[B + j]) - (i == j ? 2.0f : 0.f)) * sc;
    float cf;
    if (kind == 0) cf = sij < 0.f ? -2.0f * w : 0.f;
    else if (kind == 1) { const float rr = -sij * temperature; cf = (rr * rr > 1.0000005e-9f) ? -w / rr : 0.f; }
    else if (kind == 2 || kind == 3) cf = -w;
    else {
      const float nself = norms[tr == 0 ? i : B + j], noth = norms[tr == 0 ? B + j : i];
      const float ra = rsqrtf(nself), rb = rsqrtf(noth);
      cf = w * ra * rb;
      C2m[idx] = w * (sij * temperature) * ra * ra;
    }
    Cm[idx] = cf;
  }
}

// gradient wrt the embeddings.  which = 0: block i accumulates g_x[i] over its row of C; which = 1: block j accumulates
// g_y[j] over its row of CT.  Thread t owns coordinate e = t % D of partner group t / D (D <= 256).  Partners are taken in
// chunks of 256: the chunk's coefficients (and arg-max coordinates) go through LDS once, then every (pair, coordinate) costs a
// broadcast LDS read, one coalesced load of the partner's coordinate and 2-4 VALU instructions.  Fixed-order sums
// (deterministic).
__global__ void __launch_bounds__(256)
dib_infonce_grad_kernel(const float* __restrict__ X, const float* __restrict__ Y, const float* __restrict__ C,
                        const float* __restrict__ CT, const float* __restrict__ C2, const float* __restrict__ C2T,
                        const int* __restrict__ amax, const int* __restrict__ amaxT, int B, int D, int kind,
                        float* __restrict__ GX, float* __restrict__ GY) {
  extern __shared__ float acc[];        // [256] partial gradients
  __shared__ float cs[256], cs2[256];
  __shared__ int ams[256];
  const int which = blockIdx.y, me = blockIdx.x;
  const float* self = (which == 0 ? X : Y) + (long long)me * D;
  const float* others = which == 0 ? Y : X;
  const float* crow = (which == 0 ? C : CT) + (long long)me * B;
  const float* c2row = (which == 0 ? C2 : C2T) + (long long)me * B;
  const int* arow = (which == 0 ? amax : amaxT) + (long long)me * B;
  float* out = (which == 0 ? GX : GY) + (long long)me * D;
  const int groups = max(1, 256 / D);
  const int e = threadIdx.x % D, grp = threadIdx.x / D;
  const bool live = grp < groups;
  const float se = live ? self[e] : 0.f;
  float g = 0.f;
  for (int o0 = 0; o0 < B; o0 += 256) {
    const int n = min(256, B - o0);
    __syncthreads();
    if ((int)threadIdx.x < n) {
      cs[threadIdx.x] = crow[o0 + threadIdx.x];
      if (kind == 4) cs2[threadIdx.x] = c2row[o0 + threadIdx.x];
      if (kind == 3) ams[threadIdx.x] = arow[o0 + threadIdx.x];
    }
    __syncthreads();
    if (live) {
      for (int ol = grp; ol < n; ol += groups) {
        const float oe = others[(long long)(o0 + ol) * D + e];
        const float c = cs[ol];
        if (kind == 0 || kind == 1) {
          g += c * (se - oe);
        } else if (kind == 2) {
          const float df = se - oe;
          g += df > 0.f ? c : (df < 0.f ? -c : 0.f);
        } else if (kind == 3) {
          if (ams[ol] == e) { const float df = se - oe; g += df > 0.f ? c : (df < 0.f ? -c : 0.f); }
        } else {
          g += c * oe - cs2[ol] * se;
        }
      }
    }
  }
  acc[threadIdx.x] = live ? g : 0.f;
  __syncthreads();
  if (threadIdx.x < D) {  // fixed-order sum over the partner groups
    float s = 0.f;
    for (int q = 0; q < groups; ++q) s += acc[q * D + threadIdx.x];
    out[threadIdx.x] = s;
  }
}

// dense positional encoding of a single [N, d] matrix (reference train.py:186-188: PositionalEncoding on the Y encoder)
__global__ void __launch_bounds__(256)
dib_posenc_dense_kernel(const float* __restrict__ X, long long ldx, int n, int d, int n_blocks, float* __restrict__ P) {
  const long long total = (long long)n * d;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int r = (int)(i / d), c = (int)(i - (long long)r * d);
    const float x = X[r * ldx + c];
    float* dst = P + (long long)r * d * n_blocks + c;
    dst[0] = x;
    float fr = 2.0f;
    for (int j = 1; j < n_blocks; ++j) { dst[(long long)j * d] = sinf(fr * x); fr *= 2.0f; }
  }
}

// the same with the batch gather fused in (rows row_idx[0..n) of X; the custom loop's Y batch, train.py:226-227): n_blocks == 1
// is a plain row gather
__global__ void __launch_bounds__(256)
dib_posenc_rows_kernel(const float* __restrict__ X, long long ldx, const int* __restrict__ row_idx, int n, int d, int n_blocks,
                       float* __restrict__ P) {
  const long long total = (long long)n * d;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int r = (int)(i / d), c = (int)(i - (long long)r * d);
    const float x = X[(long long)row_idx[r] * ldx + c];
    float* dst = P + (long long)r * d * n_blocks + c;
    dst[0] = x;
    float fr = 2.0f;
    for (int j = 1; j < n_blocks; ++j) { dst[(long long)j * d] = sinf(fr * x); fr *= 2.0f; }
  }
}

// ---------------------------------------------------------------------------------------------
// Skinny output layer (out_dim <= 8, e.g. the reference's 1-unit logit, models.py:83): a [B,K] x [K,out] product is
// a GEMV-like HBM-bound stream, not an MFMA tile (N=1 padded to 64 columns wastes 98 % of the matrix core and ran
// at 0.4 TB/s).  One wave per row for fwd, elementwise dgrad, block-per-slab column reduction for wgrad.
// ---------------------------------------------------------------------------------------------
#define DIB_SKINNY_MAX 8

__global__ void __launch_bounds__(256)
dib_skinny_fwd_kernel(const float* __restrict__ A, int batch, int K, const float* __restrict__ W /*[K][out]*/,
                      const float* __restrict__ bias, int out, int act, float* __restrict__ C /*[B][out]*/) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = (gridDim.x * 256) >> 6;
  for (int b = wave; b < batch; b += nwaves) {
    const float* a = A + (long long)b * K;
    float acc[DIB_SKINNY_MAX];
#pragma unroll
    for (int o = 0; o < DIB_SKINNY_MAX; ++o) acc[o] = 0.f;
    for (int k = lane; k < K; k += 64) {
      const float av = a[k];
#pragma unroll
      for (int o = 0; o < DIB_SKINNY_MAX; ++o)
        if (o < out) acc[o] += av * W[(long long)k * out + o];
    }
#pragma unroll
    for (int o = 0; o < DIB_SKINNY_MAX; ++o) {
      if (o < out) {
        const float s = dib_wave_sum(acc[o]);
        if (lane == 0) C[(long long)b * out + o] = dib_act(act, s + (bias ? bias[o] : 0.f));
      }
    }
  }
}

// g_in[b][k] = (sum_o g[b][o] W[k][o]) * act'(a_in[b][k])
__global__ void __launch_bounds__(256)
dib_skinny_dgrad_kernel(const float* __restrict__ G /*[B][out]*/, int batch, int K, const float* __restrict__ W, int out,
                        const float* __restrict__ Ain /*[B][K] or null*/, int act, float* __restrict__ Gin) {
  const long long total = (long long)batch * K;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int b = (int)(i / K), k = (int)(i - (long long)b * K);
    float s = 0.f;
    for (int o = 0; o < out; ++o) s += G[(long long)b * out + o] * W[(long long)k * out + o];
    if (Ain != nullptr && act != 0) s *= dib_act_grad(act, Ain[i]);
    Gin[i] = s;
  }
}

// Two deterministic stages.  Stage 1: block c handles rows [c*rows_per_chunk, ...): partial[c][k*out+o] = sum_b a[b][k] g[b][o]
// and partial[c][K*out + o] = sum_b g[b][o].  Stage 2: fixed-order sum over the chunks into the gradient (slab 0).
__global__ void __launch_bounds__(256)
dib_skinny_wgrad_kernel(const float* __restrict__ A, const float* __restrict__ G, int batch, int K, int out,
                        int rows_per_chunk, float* __restrict__ partial) {
  __shared__ float red[4];
  const int r0 = blockIdx.x * rows_per_chunk, r1 = min(batch, r0 + rows_per_chunk);
  float* dst = partial + (long long)blockIdx.x * ((long long)K * out + out);
  for (int o = 0; o < out; ++o) {
    for (int k = threadIdx.x; k < K; k += 256) {
      float s = 0.f;
      for (int b = r0; b < r1; ++b) s += A[(long long)b * K + k] * G[(long long)b * out + o];
      dst[(long long)k * out + o] = s;
    }
    float sb = 0.f;
    for (int b = r0 + (int)threadIdx.x; b < r1; b += 256) sb += G[(long long)b * out + o];
    const float tot = dib_block_sum_256(sb, red);
    if (threadIdx.x == 0) dst[(long long)K * out + o] = tot;
  }
}

__global__ void __launch_bounds__(256)
dib_skinny_wgrad_reduce_kernel(const float* __restrict__ partial, int nchunks, int K, int out, float* __restrict__ dW,
                               float* __restrict__ dB) {
  __shared__ float red[4];
  const int n = K * out + out, i = blockIdx.x;  // one block per output element; fixed reduction tree => deterministic
  float s = 0.f;
  for (int c = threadIdx.x; c < nchunks; c += 256) s += partial[(long long)c * n + i];
  const float tot = dib_block_sum_256(s, red);
  if (threadIdx.x == 0) {
    if (i < K * out) dW[i] = tot;
    else dB[i - K * out] = tot;
  }
}


// ---------------------------------------------------------------------------------------------
// Fused 1-unit output head of a training step (reference models.py:83 Dense(out) + the compiled Keras loss + its part of
// tape.gradient): for every row ONE pass over the last hidden activation a[b][0..K):
//   z = a.w + bias  -> pred ;  loss / accuracy terms ;  g = dloss/dz * inv_bg -> g_pred ;
//   g_a[b][k] = g * w[k] * act'(a[b][k])            (dgrad into the last hidden layer)
//   dW[k] += a[b][k] * g ,  db += g                  (per-workgroup partials, fixed-order reduce afterwards)
// instead of four launches (skinny fwd, loss, skinny wgrad, skinny dgrad) that each re-read a or g.  One wave per row,
// a lane holds the float4 columns lane, lane + 64, ... (K <= 1024, K % 4 == 0); rows of a workgroup in a fixed order.
// partial_w: [gridDim.x][K + 1] (layout of dib_skinny_wgrad_reduce_kernel), partial_l: [gridDim.x][2] (dib_loss_finalize).
// ---------------------------------------------------------------------------------------------
template <int NC>   // float4 columns per lane = ceil(K / 256)
__global__ void __launch_bounds__(256)
dib_head_fused_kernel(int kind, const float* __restrict__ A, int batch, int K, const float* __restrict__ W,
                      const float* __restrict__ bias, const float* __restrict__ Y, long long ldy,
                      const int* __restrict__ row_idx, long long row0, float inv_bg, int act, int rows_per_block,
                      float* __restrict__ pred, float* __restrict__ g_pred, float* __restrict__ g_a,
                      float* __restrict__ partial_w, float* __restrict__ partial_l) {
  __shared__ float redw[4][1024 + 4];
  __shared__ float redl[4][2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int K4 = K >> 2;
  float4 w4[NC], pw[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int j = lane + 64 * c;
    w4[c] = j < K4 ? reinterpret_cast<const float4*>(W)[j] : make_float4(0.f, 0.f, 0.f, 0.f);
    pw[c] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const float b0 = bias ? bias[0] : 0.f;
  float pb = 0.f, lsum = 0.f, correct = 0.f;
  const int r0 = blockIdx.x * rows_per_block, r1 = min(batch, r0 + rows_per_block);
  for (int b = r0 + wave; b < r1; b += 4) {
    const float4* a = reinterpret_cast<const float4*>(A + (long long)b * K);
    float4 av[NC];
    float dot = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int j = lane + 64 * c;
      av[c] = j < K4 ? a[j] : make_float4(0.f, 0.f, 0.f, 0.f);
      dot += av[c].x * w4[c].x + av[c].y * w4[c].y + av[c].z * w4[c].z + av[c].w * w4[c].w;
    }
    const float z = dib_wave_sum(dot) + b0;
    const long long row = row_idx ? (long long)row_idx[b] : row0 + b;
    const float yy = Y[row * ldy];
    float l, gg;
    if (kind == 0) {  // Keras BinaryCrossentropy(from_logits=True), same expressions as dib_loss_kernel
      l = fmaxf(z, 0.f) - z * yy + log1pf(expf(-fabsf(z)));
      gg = 1.0f / (1.0f + expf(-z)) - yy;
    } else {          // 'mse'
      const float d = z - yy;
      l = d * d;
      gg = 2.f * d;
    }
    gg *= inv_bg;
    if (lane == 0) {
      pred[b] = z;
      if (g_pred != nullptr) g_pred[b] = gg;
      lsum += l;
      correct += ((z > 0.5f ? 1.f : 0.f) == yy) ? 1.f : 0.f;
      pb += gg;
    }
    if (g_a == nullptr) continue;   // validation: prediction + loss terms only (workgroup-uniform)
    float4* ga = reinterpret_cast<float4*>(g_a + (long long)b * K);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int j = lane + 64 * c;
      if (j < K4) {
        ga[j] = make_float4(gg * w4[c].x * dib_act_grad(act, av[c].x), gg * w4[c].y * dib_act_grad(act, av[c].y),
                            gg * w4[c].z * dib_act_grad(act, av[c].z), gg * w4[c].w * dib_act_grad(act, av[c].w));
        pw[c].x += av[c].x * gg; pw[c].y += av[c].y * gg; pw[c].z += av[c].z * gg; pw[c].w += av[c].w * gg;
      }
    }
  }
  // fixed-order sum of the four waves' partials
  if (g_a != nullptr) {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int j = lane + 64 * c;
      if (j < K4) *reinterpret_cast<float4*>(&redw[wave][4 * j]) = pw[c];
    }
  }
  if (lane == 0) { redw[wave][K] = pb; redl[wave][0] = lsum; redl[wave][1] = correct; }
  __syncthreads();
  if (g_a != nullptr) {
    float* dst = partial_w + (long long)blockIdx.x * (K + 1);
    for (int i = threadIdx.x; i <= K; i += 256) dst[i] = redw[0][i] + redw[1][i] + redw[2][i] + redw[3][i];
  }
  if (threadIdx.x < 2) partial_l[2 * blockIdx.x + threadIdx.x] = redl[0][threadIdx.x] + redl[1][threadIdx.x] + redl[2][threadIdx.x] + redl[3][threadIdx.x];
}
