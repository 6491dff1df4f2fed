// dib_api.hip - C ABI (include/dib_hip.h) of the MI355X Distributed-IB hot path: layout, workspace
// carving and the launch sequences of the forward / backward / optimizer steps.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <vector>

#include "../../include/dib_hip.h"
#include "dib_elementwise.h"
#include "dib_gemm.h"
#include "dib_infonce_mfma.h"
#include "dib_fused.h"
#include "dib_tail.h"
#include "dib_small.h"
#include "dib_st.h"
#include "dib_attn.h"
#include "dib_attn_small.h"
#include "dib_st_chain.h"
#include "../../include/dib_st.h"

// every kernel launch of the library goes through this macro: dib_launch_count() reports how many a step issues (bench.py).
// Relaxed atomic: entry points may run on several host threads at once (include/dib_hip.h "Threads").
static std::atomic<unsigned long long> g_dib_launches{0};
#define DIB_LAUNCH(...) do { g_dib_launches.fetch_add(1, std::memory_order_relaxed); hipLaunchKernelGGL(__VA_ARGS__); } while (0)

namespace {

constexpr int64_t kAlign = 64;  // floats (256 B)
constexpr int kMaxSplits = 32;   // partial slabs of a split-batch weight gradient
// Row-tile kernels of dib_small.h: hard limits (they size workspace regions); WHICH batches take them is the "small_wgs" rule
constexpr int kSmallMaxBatch = 2048;     // rows
constexpr int kSmallMaxEncWgs = 1024;    // row tiles x features (d(W1|b1) partials: one [16][H1] block per encoder workgroup)
constexpr int kSplitRows = 512;  // minimum batch rows per wgrad split: 8 K-tiles of 64 (measured: 2048 left mid-size batches with 16-256 workgroups)
inline int64_t align_up(int64_t v, int64_t a = kAlign) { return (v + a - 1) / a * a; }
inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

struct GemmCall {  // one grouped launch: slice [first, first+count) of the descriptor table
  int first = 0, count = 0;
  int max_m = 0, max_n = 0;  // max logical dims over the groups (-1 => batch)
};

}  // namespace

static int wgrad_max_splits();   // dib_set_tuning("wgrad_max_splits"): defined with the tuning table below

struct dib_layout {
  int F = 0, n_enc = 0, E = 0, n_int = 0, out_dim = 0, use_pe = 0, n_freq = 0, act = 0, out_act = 0;
  std::vector<int> dims, enc_units, int_units;
  int sum_d = 0, pw = 0, n_blocks = 1;    // pw = total encoder-input width, n_blocks = 1 + #sinusoids
  std::vector<int> in_dim, in_off, x_off; // per feature: encoder input width / its first column in P / in x
  std::vector<int> enc_width;             // [n_enc+1] per-feature output width of each encoder layer
  std::vector<int> int_width;             // [n_int+1]
  int64_t n_params = 0;
  long long max_wgrad_tiles64 = 0;        // most 64 x 64 output tiles any one weight-gradient launch has (all groups)
  std::vector<std::vector<int64_t>> enc_w_off, enc_b_off;  // [layer][feature]
  std::vector<int64_t> int_w_off, int_b_off;
  // descriptor table
  std::vector<DibGemmGroup> table;
  std::vector<int4> colmap;
  std::vector<GemmCall> enc_fwd, enc_dgrad, enc_wgrad, int_fwd, int_dgrad, int_wgrad;
  const DibGemmGroup* dev_groups = nullptr;
  const int4* dev_colmap = nullptr;
  // fused encoder-bank kernels (dib_fused.h): -1 = not applicable, else index into the instantiation table
  int fused_id = -1;
  std::vector<long long> fused_offs;   // [3][F] kernel offsets then [3][F] bias offsets
  std::vector<int4> featmap;           // [F] {d_f, in_dim_f, x column, 0}
  const long long* dev_fused_offs = nullptr;
  const int4* dev_featmap = nullptr;
  const unsigned* step_dev = nullptr;  // optional device-resident noise step (dib_layout_set_step_counter)
  // small-batch row-tile kernels (dib_small.h): which halves of the network they cover for this architecture
  bool sb_enc = false, sb_int = false;
  int sb_int_lds = 0;                  // dynamic LDS bytes of dib_small_integration_kernel
  // merged weight-gradient table (one per batch size, kept alive for the asynchronous upload of dib_workspace_init):
  // groups [0, n_enc F): encoder layers 1 .. n_enc (feature-major), then the integration layers 0 .. n_int
  int wg_groups() const { return n_enc * F + n_int + 1; }
  mutable std::map<int, std::vector<DibGemmGroup>> wg_tables;
  mutable std::mutex wg_mu;            // two threads may initialise workspaces of one layout (include/dib_hip.h "Threads")
  // Most batch slabs any weight-gradient launch of this layout has written per (batch size, parameter block) - see
  // retire_stale_slabs.  Conservative across the layout's workspaces (guarded by wg_mu).
  mutable std::map<std::pair<int, long long>, int> slab_hwm;

  // ---- workspace map (float offsets), all per-row widths scale with the batch ----
  struct WsMap {
    int64_t P, enc_out, U, pred, g_pred, g_u, dout;
    std::vector<int64_t> enc_h, int_h, g_enc_h, g_int_h;
    int64_t step_out, kl_partial, loss_partial, wgrad_partial, dw1_partial, h2mask, h1mask, skinny_partial, sync, wg_table, total;
    int64_t cl_sync; std::vector<int64_t> cl_x;   // cluster mode of the row-tile integration kernel (dib_small.h)
    int skinny_chunks, skinny_rows;
    int kl_blocks, loss_blocks, nsplit, rows_per_split;
  };
  WsMap map(int B) const {
    WsMap m;
    int64_t o = 0;
    auto take = [&](int64_t nfloats) { int64_t r = o; o = align_up(o + nfloats); return r; };
    m.P = take((int64_t)B * pw);
    for (int l = 0; l < n_enc; ++l) m.enc_h.push_back(take((int64_t)B * F * enc_units[l]));
    m.enc_out = take((int64_t)B * F * 2 * E);
    m.U = take((int64_t)B * F * E);
    for (int l = 0; l < n_int; ++l) m.int_h.push_back(take((int64_t)B * int_units[l]));
    m.pred = take((int64_t)B * out_dim);
    m.g_pred = take((int64_t)B * out_dim);
    for (int l = 0; l < n_int; ++l) m.g_int_h.push_back(take((int64_t)B * int_units[l]));
    m.g_u = take((int64_t)B * F * E);
    m.dout = take((int64_t)B * F * 2 * E);
    for (int l = 0; l < n_enc; ++l) m.g_enc_h.push_back(take((int64_t)B * F * enc_units[l]));
    m.step_out = take(F + 3);
    const int E4 = (E + 3) / 4;
    const int rpb = std::max(1, 256 / E4);
    m.kl_blocks = cdiv(B, rpb);
    m.kl_partial = take((int64_t)std::max(m.kl_blocks, 8 * 256) * F);  // fused fwd: one row per wave of the persistent grid
    m.loss_blocks = cdiv(B, 256);
    m.loss_partial = take((int64_t)std::max(m.loss_blocks, 512) * 2);  // also the fused output head's per-workgroup partials (<= 512)
    // split-batch wgrad: rows_per_split multiple of 32, <= 32 splits, >= kSplitRows rows per split ...
    int ns = std::min(std::min(kMaxSplits, wgrad_max_splits()), std::max(1, B / kSplitRows));
    // ... unless the layout is so narrow that even its largest weight gradient stays under one workgroup per CU with that
    // many splits (BASELINE config 2, the pendulum layout [2,1,2,1]: 16 tiles x 4 splits of 512 rows at B = 2048 - five
    // launches of 18-24 us, each a workgroup walking 16 dependent K-tiles, 100 of the 510 us step,
    // profiles/r04i_config2_loop_kernel_stats_b2048.csv): then slabs of >= 128 rows
    if (B >= 256 && max_wgrad_tiles64 * ns < 256) ns = std::min(std::min(kMaxSplits, wgrad_max_splits()), std::max(ns, B / 128));
    int rps = cdiv(cdiv(B, ns), 32) * 32;
    ns = cdiv(B, rps);
    m.nsplit = ns;
    m.rows_per_split = rps;
    m.wgrad_partial = take(ns > 1 ? (int64_t)ns * align_up(n_params, 4) : 0);
    // fused backward: per-wave partials of d(W1|b1), [<= ceil(256/F) workgroups x 8 waves][F][16][H1]
    // (small-batch path: one partial per 16-row tile, [<= kSmallMaxEncWgs (tile, feature) pairs][16][H1])
    m.dw1_partial = take(std::max<int64_t>(fused_id >= 0 && n_enc == 2 ? (int64_t)cdiv(256, F) * 8 * F * 16 * enc_units[0] : 0,
                                           sb_enc && B <= kSmallMaxBatch && (int64_t)cdiv(B, DIB_SMALL_ROWS) * F <= kSmallMaxEncWgs
                                               ? (int64_t)cdiv(B, DIB_SMALL_ROWS) * F * 16 * enc_units[0] : 0));
    // [F][B][2] x 64-bit act' masks (fused fwd -> fused bwd), one bit per hidden unit
    m.h2mask = take(fused_id >= 0 ? (int64_t)F * B * 4 : 0);
    m.h1mask = take(fused_id >= 0 ? (int64_t)F * B * 4 : 0);
    // skinny output layer wgrad: row chunks of >= 64 rows (>= 16 up to B = 2048), <= 512 chunks
    // (16-row chunks for small batches: with 64 the fused output head of the reference's default B = 128 step ran on 2
    // workgroups, each wave walking 16 rows one after the other - 21 us, profiles/r04l_default_batch_kernel_stats.csv)
    m.skinny_rows = std::max(B <= 2048 ? 16 : 64, cdiv(B, 512));
    m.skinny_chunks = cdiv(B, m.skinny_rows);
    {
      const int win = n_int == 0 ? F * E : int_units[n_int - 1];
      m.skinny_partial = take(out_dim <= 8 ? (int64_t)m.skinny_chunks * ((int64_t)win * out_dim + out_dim) : 0);
    }
    m.sync = take(DIB_TAIL_SYNC_WORDS);   // arrival counters of dib_step_tail (zeroed by dib_workspace_init, self-cleaning)
    // descriptors of ALL weight gradients of a step with absolute workspace offsets for THIS batch size (written by
    // dib_workspace_init): one grouped launch instead of one per layer (merged_wgrad)
    m.wg_table = take((int64_t)wg_groups() * (int64_t)(sizeof(DibGemmGroup) / sizeof(float)));
    // cluster mode of the row-tile integration kernel: arrival counters per row tile (zeroed by dib_workspace_init, self-cleaning)
    // and the hidden activations' exchange buffers of launches that write no stashes
    const bool cl_ok = sb_int && B <= kSmallMaxBatch;
    m.cl_sync = take(cl_ok ? 2ll * cdiv(B, DIB_SMALL_ROWS) * DIB_SMALL_CL_SYNC_WORDS : 0);   // x 2: a companion network's (paired grid)
    for (int l = 0; l < n_int; ++l) m.cl_x.push_back(take(cl_ok ? (int64_t)B * int_units[l] : 0));
    m.total = o;
    return m;
  }
};

namespace {

// ---- optional live kernel timing (bench.py roofline): HIP events around every launch, on the launch stream ----
// categories = kernel symbols: 0..11 dib_gemm_kernel<MODE,NI,NJ> at MODE*4 + (NI-1)*2 + (NJ-1); 12 fused encoder fwd;
// 13 fused encoder bwd; 14 every other (HBM-bound) kernel; 15 dib_attn_fwd_kernel; 16 dib_attn_bwd_kernel
constexpr int kProfCats = 17;
constexpr int kProfFusedFwd = 12, kProfFusedBwd = 13, kProfOther = 14, kProfAttnFwd = 15, kProfAttnBwd = 16;
struct Prof {   // diagnostics (bench.py roofline): the tables are guarded, so a second thread's launches are recorded, not racy
  std::atomic<bool> on{false};
  std::mutex mu;
  std::vector<hipEvent_t> pool;                     // recycled events
  std::vector<std::pair<hipEvent_t, hipEvent_t>> spans[kProfCats];
  hipEvent_t get() {
    if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
  }
} g_prof;

struct ProfScope {
  int cat; hipStream_t st; hipEvent_t a = nullptr, b = nullptr;
  ProfScope(int c, hipStream_t s) : cat(c), st(s) {
    // the small HBM-bound kernels are not bracketed (event pairs serialise kernel boundaries: ~10 us each); rocprofv3
    // reports them (profiles/*_kernel_stats.csv)
    if (g_prof.on.load(std::memory_order_relaxed) && cat != 14) {
      { std::lock_guard<std::mutex> lk(g_prof.mu); a = g_prof.get(); b = g_prof.get(); }
      (void)hipEventRecord(a, st);
    }
  }
  ~ProfScope() {
    if (a) { (void)hipEventRecord(b, st); std::lock_guard<std::mutex> lk(g_prof.mu); g_prof.spans[cat].push_back({a, b}); }
  }
};

const char* kVersion = "dib_hip 0.4 (gfx950: fused encoder-bank fwd/bwd + grouped fp32-MFMA GEMM + flash attention)";

int act_ok(int a) { return a >= 0 && a <= 7; }

// Off = {fixed element offset, offset per batch row} : activations are feature-major [F][B][width]
struct Off { int64_t fixed = 0, per_batch = 0; };
inline Off fixed_off(int64_t o) { Off r; r.fixed = o; return r; }
inline Off batch_off(int64_t o) { Off r; r.per_batch = o; return r; }

DibGemmGroup make_group(Off a, int lda, Off b, int ldb, Off c, int ldc, int64_t bias_off, Off aux, int ldaux, int M,
                        int N, int K) {
  DibGemmGroup g;
  std::memset(&g, 0, sizeof(g));
  g.a_off = a.fixed; g.a_boff = a.per_batch; g.b_off = b.fixed; g.b_boff = b.per_batch;
  g.c_off = c.fixed; g.c_boff = c.per_batch; g.aux_off = aux.fixed; g.aux_boff = aux.per_batch;
  g.bias_off = bias_off;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ldaux = ldaux;
  return g;
}

// Tile / split rules.  The defaults are the measured choices; dib_set_tuning (include/dib_hip.h) is the ONE documented way to
// change them (A/B measurements, tools/ab_bench.sh) - the library reads no environment variable.
struct Tuning {
  int fwd_small_wgs = 512;   // forward/dgrad: below this many 128-row workgroups use 64-row tiles
  int fwd_narrow_wgs = 1024; // forward: below this many 64x128 workgroups use 64x64 tiles (round 3: 512 -> 1024, the set
                             // transformer's q/k/v projection at 1600 tokens: step 1.99 -> 1.87 ms; profiles/r03l_forward_tile_rule.txt)
  int stream_rows = 8192;    // GEMMs with at least this many streamed rows load / store them non-temporally (1 << 30: never)
  int split_policy = 1;      // weight gradients of the layout: 1 = pick the batch-split count per launch so that the workgroups
                             // fill whole rounds of the chip's workgroup slots (pick_wgrad_splits); 0 = the layout-wide count
  int split_overhead = 128;  // ... with this per-workgroup fixed cost, in batch rows (prologue + partial-tile store)
  int fused_encoder = 1;     // layouts created from now on may use the fused encoder-bank kernels (0: grouped-GEMM path)
  int fused_head = 1;        // dib_output_head_fused_supported may answer 1
  int small_batch = 1;       // row-tile kernels (csrc/dib_small.h) where the layout allows, while ...
  int small_wgs = 512;       // ... (row tiles of 16) x (features) <= this (and batch <= 2048)
  int mlp_row_tiles = 1;     // ... and for a plain MLP (dib_mlp_small_*: the custom loop's output encoder)
  int infonce_one_launch = 1; // dib_infonce_fwd_bwd at B <= 128, D <= 64 (dot-product similarities): one launch instead of three
  int attn_small_bwd_waves = 8;  // dib_attention_bwd for <= 64 particles: 8 waves (two per SIMD) or the 4-wave kernel
  int wgrad_flat_tile = 1;   // weight gradients with <= 32 rows and >= 256 columns on the 32 x 256 tile (0: 64 x 128, A/B)
  int attn_fwd_waves = 8;    // dib_attention_fwd for P >= 256: 8-wave workgroups of 256 queries sharing one staged K / V tile (4: the 4-wave
                             // kernel, which shorter sets always take; bit-identical outputs)
  int int_cluster_short_exchange = 1;  // clusters on one XCD exchange through that XCD's L2 (0: always the agent-scope protocol - the
                             // path a cluster takes when it is NOT on one XCD; tests)
  int int_cluster = 8;       // row-tile integration kernel: workgroups per row tile (each a column slice of every layer, exchange
                             // through L2: dib_small.h "cluster mode"; <= 1: one per tile) while row tiles x this <= ...
  int int_cluster_wgs = 256; // ... this (one workgroup per CU; 8 per tile up to 32 row tiles, 4 up to 64: profiles/r06u_int_cluster_sweep.txt) and
  int int_cluster_min_weights = 65536;  // ... the network's hidden layers have at least this many weights (measured down to 4
                             // features x 32 -> 256 -> 256: 98 304)
  int wgrad_max_splits = 32; // most batch slabs of a layout's weight gradients (<= 32; read when a workspace is sized: set it first)
  int num_cus = 0;           // compute units the split rule prices rounds with; 0 = the current device's own count (device_cus)
};
// Process-wide and written ONLY by dib_set_tuning, which the header documents as a configuration call made while no other
// entry point is running; every other entry point only reads it.
inline Tuning& tuning() { static Tuning t; return t; }
inline const Tuning& knobs() { return tuning(); }
}  // namespace
static int wgrad_max_splits() { return std::max(1, std::min(32, knobs().wgrad_max_splits)); }
namespace {
// compute units of the CURRENT device, queried once per device ordinal (no process-wide "the device": one process may drive
// several GPUs from several threads)
inline int device_cus() {
  static std::atomic<int> cus[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  int v = cus[dev].load(std::memory_order_relaxed);
  if (v > 0) return v;
  if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) return 256;
  cus[dev].store(v, std::memory_order_relaxed);
  return v;
}
inline int split_rule_cus() { return knobs().num_cus > 0 ? knobs().num_cus : device_cus(); }

template <int MODE, int NI, int NJ, bool FLAT = false>
int launch_gemm_t(const DibGemmGroup* dev_groups, const GemmCall& c, int M, int N, const float* A, const float* B, float* C,
                  const float* bias, const float* aux, float* bias_out, int batch, int act, int nsplit,
                  int rows_per_split, long long split_stride, hipStream_t st) {
  const int tm = cdiv(M, FLAT ? 32 : 64 * NI), tn = cdiv(N, FLAT ? 256 : 64 * NJ);
  // grid.y / grid.z are limited to 65535: a clean return code instead of a launch error
  if (c.count > 65535 || (MODE == 2 && (long long)tm * tn > 65535)) return DIB_E_UNSUPPORTED;
  dim3 grid;
  if (MODE == 2) grid = dim3(nsplit, tm * tn, c.count);
  else grid = dim3(8 * cdiv(tm, 8) * tn, 1, c.count);  // XCD-aware 1-D tile order, see dib_gemm.h
  // K-tile depth per tile shape (each a same-box A/B, profiles/HISTORY.md): 64 for the 128 x 128 tile of every mode - one
  // prefetch + barrier pair per 64-deep MFMA phase hides the HBM latency a 32-deep phase exposes (+18 %) - and for the 64 x 128
  // weight-gradient tile of the 256 x 256 integration layer (0.136 -> 0.124 ms); 32 for the other narrow tiles.
  // (the flat 32 x 256 weight-gradient tile: 32 - its 256-column operand tile at 64 deep would need 76 KB of static LDS)
  constexpr int BK = FLAT ? 32 : ((NI == 2 && NJ == 2) ? 64 : ((MODE == 2 && NI == 1 && NJ == 2) ? 64 : 32));
  // cache policy of the streamed operands / outputs (dib_gemm.h: stream_flags): non-temporal from 8192 streamed rows up
  // (DIB_GEMM_STREAM_ROWS; M for forward / dgrad, the contracted rows for a weight gradient)
  const long long streamed_rows = MODE == 2 ? (long long)nsplit * rows_per_split : (long long)M;
  // ... and the output non-temporally only when it cannot stay in the 256 MB infinity cache for its consumer anyway (the 67 MB
  // hidden activation of the integration network, stored non-temporally, cost the fused head that reads it next 19 us)
  const bool big_out = MODE != 2 && (long long)M * N * (long long)sizeof(float) * c.count >= (256ll << 20);
  const int stream_flags = streamed_rows >= knobs().stream_rows ? (big_out ? 3 : 1) : 0;
  DIB_LAUNCH((dib_gemm_kernel<MODE, NI, NJ, BK, FLAT>), grid, dim3(256), 0, st, dev_groups + c.first, A, B, C,
                     bias, aux, bias_out, batch, act, tm, tn, rows_per_split, split_stride, stream_flags);
  return (int)hipGetLastError();
}


// Batch-split count of one weight-gradient launch: `tiles` output tiles (all groups) x ns splits of rps batch rows on `slots`
// co-resident workgroup slots (256 CUs x workgroups per CU of the tile shape).  Equal-length workgroups execute in
// ceil(tiles ns / slots) rounds, so the launch takes ~ rounds x (rps + a fixed cost per workgroup).  The layout-wide rule - 32
// splits of 2048 rows at B = 65536 - is exact for F = 64 (64 tiles x 32 = 4.0 rounds of 512) and off for F = 50: 1600
// workgroups = 3.1 rounds, the fourth 1/8 full; the narrow last-layer gradient (4 workgroups per CU) with its splits halved
// ran 800 workgroups of 4096 rows where 1000 of 3328 fit one round.  BASELINE config 4: encoder wgrads at 0.55-0.61 of the
// fp32-MFMA peak against 0.70-0.72 for config 3 (profiles/r04a_config4_*).  Candidates: 1 .. max_splits splits of a multiple of
// 64 rows (whole K-tiles), at least kSplitRows rows; the cheapest wins, ties go to FEWER, longer workgroups.
// Measured (F = 50, B = 65536, ms/step with one launch's count forced, profiles/r04d_split_sweep_F50.txt): encoder layers 2+3,
// 50 tiles: 32 splits 6.80, 30: 6.78, 28: 6.87, 25: 6.74, 20: 6.68 (2 full rounds), 16: 6.99, 10: 6.71 (1 round);
// integration layer 1, 26 tiles: 32: 6.80, 29: 6.73, 24: 6.83, 19: 6.66 (1 round), 16: 6.82, 13: 7.01.  (A second model, "what
// a CU executes is serial: ceil(tiles ns / 256) x rps", picked 25 and 29 there and measured no gain: r04c.)
// Slabs beyond the chosen count are never written by this launch and stay zero (include/dib_hip.h workspace contract).
static void pick_wgrad_splits(long long tiles, int slots, int K, int max_splits, int* ns_out, int* rps_out) {
  const int min_rows = std::min(kSplitRows, std::max(64, *rps_out));   // narrow layouts come in with shorter slabs (WsMap)
  // The caller's (layout-wide) split is kept whenever it fills its rounds to at least 85 %: BASELINE config 3 (F = 64: 64 or 32
  // tiles x 32 or 16 splits = whole rounds at every batch size) then runs exactly the launches rounds 2-3 measured and
  // validated (same-box A/B of an unconditional rule vs no rule there: 8.00-8.08 vs 7.99-8.02 ms/step,
  // profiles/r04e_split_policy_final_ab.txt).
  const auto cost_of = [&](int ns, int rps) {
    return (double)((tiles * ns + slots - 1) / slots) * (rps + knobs().split_overhead);
  };
  {
    const long long wgs = tiles * *ns_out, rounds = (wgs + slots - 1) / slots;
    if ((double)wgs >= 0.85 * (double)(rounds * slots)) return;
  }
  double best = 1e300;
  int bns = *ns_out, brps = *rps_out;
  for (int ns = 1; ns <= max_splits; ++ns) {
    const int rps = cdiv(cdiv(K, ns), 64) * 64;
    if (ns > 1 && rps < min_rows) break;
    if (cdiv(K, rps) != ns) continue;   // the same split as a smaller ns
    const double cost = cost_of(ns, rps);
    if (cost < best) {
      best = cost;
      bns = ns;
      brps = rps;
    }
  }
  if (best >= cost_of(*ns_out, *rps_out)) return;
  *ns_out = bns;
  *rps_out = brps;
}

template <int MODE>
int launch_gemm(const DibGemmGroup* dev_groups, const GemmCall& c, const float* A, const float* B, float* C,
                const float* bias, const float* aux, float* bias_out, int batch, int act, int nsplit, int rows_per_split,
                long long split_stride, hipStream_t st, bool auto_split = false, int max_splits = 0, int* ns_used = nullptr) {
  if (ns_used) *ns_used = nsplit;
  if (c.count == 0) return DIB_OK;
  const int M = c.max_m < 0 ? batch : c.max_m;
  const int N = c.max_n < 0 ? batch : c.max_n;
  bool ni1 = (MODE == 2) && M <= 64, nj1 = N <= 64;   // narrow tiles for narrow outputs
  if (MODE != 2) {
    // few 128-row tiles (small batches): 64-row tiles double the workgroup count (2 fit per CU at 128x128, 4 at 64x128)
    const long long wgs = (long long)cdiv(M, 128) * cdiv(N, nj1 ? 64 : 128) * c.count;
    if (wgs < knobs().fwd_small_wgs) ni1 = true;
    // still under two workgroups per CU: halve the per-wave work once more.  Forward GEMMs switch below 1024 workgroups
    // (measured at B = 8192: the integration forward on 512 64x64 tiles instead of 256 64x128 tiles, step -30 us); the
    // dgrads measured no different and keep the round-1 threshold.
    if (ni1 && !nj1 && (long long)cdiv(M, 64) * cdiv(N, 128) * c.count < (MODE == 0 ? knobs().fwd_narrow_wgs : 128)) nj1 = true;
  }
  if (MODE == 2 && !ni1 && !nj1) {
    // small weight gradients (e.g. a 256x256 layer): 128x128 tiles x splits do not fill 256 CUs -> 64-row tiles
    const long long wgs = (long long)cdiv(M, 128) * cdiv(N, 128) * nsplit * c.count;
    if (wgs < 256) ni1 = true;
    if (ni1 && !nj1 && (long long)cdiv(M, 64) * cdiv(N, 128) * nsplit * c.count < 128) nj1 = true;  // tiny batches
  }
  if (MODE == 2 && auto_split && nsplit > 1 && knobs().split_policy) {
    // co-resident workgroups per CU of each tile shape (LDS / register budget of dib_gemm_kernel<2, NI, NJ, BK>)
    const int per_cu = (!ni1 && !nj1) ? 2 : ((!ni1 && nj1) ? 4 : (ni1 && !nj1) ? 3 : 4);
    const long long tiles = (long long)cdiv(M, ni1 ? 64 : 128) * cdiv(N, nj1 ? 64 : 128) * c.count;
    pick_wgrad_splits(tiles, split_rule_cus() * per_cu, batch, std::max(nsplit, max_splits), &nsplit, &rows_per_split);
    if (ns_used) *ns_used = nsplit;
  }
  ProfScope ps(MODE * 4 + (ni1 ? 0 : 2) + (nj1 ? 0 : 1), st);
  if constexpr (MODE == 2) {
    // a 32-row operand against a wide one (q / k / v weight gradients of the set transformer): the flat 32 x 256 tile
    if (M <= 32 && N >= 256 && knobs().wgrad_flat_tile)
      return launch_gemm_t<2, 1, 2, true>(dev_groups, c, M, N, A, B, C, bias, aux, bias_out, batch, act, nsplit, rows_per_split,
                                          split_stride, st);
  }
#define DIB_GO(NI, NJ) launch_gemm_t<MODE, NI, NJ>(dev_groups, c, M, N, A, B, C, bias, aux, bias_out, batch, act, nsplit, \
                                                   rows_per_split, split_stride, st)
  if (ni1) return nj1 ? DIB_GO(1, 1) : DIB_GO(1, 2);
  return nj1 ? DIB_GO(2, 1) : DIB_GO(2, 2);
#undef DIB_GO
}

__global__ void dib_write_desc_kernel(DibGemmGroup* dst, DibGemmGroup g) { *dst = g; }

// The reducers (dib_grads_finalize, the step tail) sum ALL the workspace's slabs of every parameter block, and a launch that
// chose `ns` splits writes slabs [0, ns) of its blocks: the slabs above stay as they are - zero since dib_workspace_init unless a
// DIFFERENT launch over the same block chose more splits earlier (the split rule prices whole launches: the row-tile regime's one
// grouped launch of all weight gradients picked 3 slabs at B = 512 where the per-layer launches of dib_integration_bwd /
// dib_encoder_bank_bwd - the custom-loss entry of a 1-unit output - picked 4, and a training step after a custom-loss step summed
// that step's fourth slab into its gradients).  Per (batch, block) the layout remembers the most slabs any launch has written;
// a launch that writes fewer zero-fills the difference behind itself.  Programs that stay on one path never pay; a program that
// alternates pays a few small memsets per step.  A launch being CAPTURED into a hipGraph cannot know what will run between its
// replays: it zero-fills every slab it does not write (slab_count = the slabs the workspace holds).
static int retire_stale_slabs(const dib_layout* l, int batch, const DibGemmGroup* host_groups, int count, int ns, int slab_count,
                              float* gt, long long split_stride, hipStream_t st) {
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  const bool capturing = hipStreamIsCapturing(st, &cap) == hipSuccess && cap == hipStreamCaptureStatusActive;
  std::lock_guard<std::mutex> lk(l->wg_mu);
  for (int i = 0; i < count; ++i) {
    const DibGemmGroup& g = host_groups[i];
    int& mark = l->slab_hwm[std::make_pair(batch, g.c_off)];
    mark = std::max(mark, ns);
    const int hwm = capturing ? std::max(mark, slab_count) : mark;
    if (ns >= hwm) continue;
    const size_t wbytes = (size_t)g.M * (size_t)g.ldc * sizeof(float);   // the block's rows are contiguous (ldc == N)
    for (int s = ns; s < hwm; ++s) {
      hipError_t e = hipMemsetAsync(gt + (long long)s * split_stride + g.c_off, 0, wbytes, st);
      if (e == hipSuccess && g.bias_off >= 0)
        e = hipMemsetAsync(gt + (long long)s * split_stride + g.bias_off, 0, (size_t)g.N * sizeof(float), st);
      if (e != hipSuccess) return (int)e;
    }
  }
  return DIB_OK;
}

template <int MODE>
int launch_gemm(const dib_layout* l, const GemmCall& c, const float* A, const float* B, float* C, const float* bias,
                const float* aux, float* bias_out, int batch, int act, int nsplit, int rows_per_split,
                long long split_stride, hipStream_t st, int slab_count = 0) {
  // the layout's weight gradients contract over the batch: their split count is chosen per launch (pick_wgrad_splits) among
  // 1 .. slab_count (the partial slabs the workspace holds)
  int ns_used = nsplit;
  int rc = launch_gemm<MODE>(l->dev_groups, c, A, B, C, bias, aux, bias_out, batch, act, nsplit, rows_per_split, split_stride,
                             st, /*auto_split=*/MODE == 2, slab_count, &ns_used);
  if (MODE == 2 && rc == DIB_OK && slab_count > 1)
    rc = retire_stale_slabs(l, batch, l->table.data() + c.first, c.count, ns_used, slab_count, C, split_stride, st);
  return rc;
}

inline int grid_for(int64_t n, int per_block = 256, int cap = 256 * 16) {
  return (int)std::max<int64_t>(1, std::min<int64_t>((n + per_block - 1) / per_block, cap));
}

}  // namespace


// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute: track it per device ordinal (two engines on two
// GPUs in one process are allowed).
// Two host threads may reach the same first launch together (include/dib_hip.h "Threads"): the flag is published only AFTER the
// attribute call (release in the destructor), the slow path is serialised, the steady state is one acquire load.
static std::mutex g_attr_mu;
struct AttrOnce {
  std::atomic<bool>* slot = nullptr;
  bool need = false;
  std::unique_lock<std::mutex> lk;
  explicit AttrOnce(std::atomic<bool> (&done)[64]) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { need = true; return; }
    slot = &done[dev];
    if (slot->load(std::memory_order_acquire)) return;
    lk = std::unique_lock<std::mutex>(g_attr_mu);
    need = !slot->load(std::memory_order_relaxed);
  }
  explicit operator bool() const { return need; }
  ~AttrOnce() { if (need && slot) slot->store(true, std::memory_order_release); }
};

template <int H1, int H2, int E, bool RELU>
static int launch_fused_fwd(const DibFusedFwdArgs& a, int gx, int F, hipStream_t st) {
  using C = DibFusedCfg<H1, H2, E>;
  const size_t lds = (size_t)C::LDS_FLOATS * sizeof(float);
  static std::atomic<bool> attr_set[64];
  if (AttrOnce once(attr_set); once) {
    hipError_t e = hipFuncSetAttribute((const void*)dib_fused_encoder_fwd_kernel<H1, H2, E, RELU>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  DIB_LAUNCH((dib_fused_encoder_fwd_kernel<H1, H2, E, RELU>), dim3(gx, F), dim3(512), lds, st, a);
  return (int)hipGetLastError();
}

// Persistent grid of the fused kernels: gx workgroups per feature, each looping over batch tiles.  One workgroup fills a
// CU (150 KB of LDS), so gx = floor(256 / F): the whole grid is co-resident.  (Rounding UP - the round-1 rule - gave F = 50
// 6 x 50 = 300 workgroups on 256 CUs: a second, 17 %-full wave of workgroups doubled the kernel time.)
static int fused_gx(const dib_layout* l, int batch) { return std::max(1, std::min(cdiv(batch, 256), std::max(1, 256 / l->F))); }

static bool fused_bwd_ok(const dib_layout* l);

static int fused_encoder_fwd(dib_layout* l, const dib_layout::WsMap& m, float* w, const float* x, int64_t ldx,
                             const int32_t* row_idx, int64_t row0, int batch, const float* params, uint64_t seed,
                             uint32_t step, int deterministic, hipStream_t st, int* gx_out) {
  DibFusedFwdArgs a;
  a.P = w + m.P; a.row_idx = (const int*)row_idx; a.row0 = row0; a.batch = batch; a.params = params;
  a.w_off = l->dev_fused_offs; a.b_off = l->dev_fused_offs + 3 * l->F; a.featmap = l->dev_featmap;
  a.n_blocks = l->n_blocks; a.act = l->act;
  a.h1 = w + m.enc_h[0]; a.h2 = w + m.enc_h[1]; a.enc_out = w + m.enc_out; a.U = w + m.U;
  a.kl_partial = w + m.kl_partial; a.F = l->F; a.seed = seed; a.step = step;
  a.deterministic = deterministic & DIB_FWD_DETERMINISTIC;
  a.h2mask = (unsigned long long*)(w + m.h2mask);
  a.h1mask = fused_bwd_ok(l) ? (unsigned long long*)(w + m.h1mask) : nullptr;
  if (deterministic & DIB_FWD_INFERENCE) { a.h1 = nullptr; a.h2 = nullptr; a.h2mask = nullptr; a.h1mask = nullptr; }  // no backward follows
  a.step_dev = l->step_dev;
  const int gx = fused_gx(l, batch);
  *gx_out = gx;
  ProfScope ps(kProfFusedFwd, st);
  switch (l->fused_id) {
    case 0: return l->act == 1 ? launch_fused_fwd<128, 128, 32, true>(a, gx, l->F, st)
                                : launch_fused_fwd<128, 128, 32, false>(a, gx, l->F, st);
    case 1: return l->act == 1 ? launch_fused_fwd<32, 32, 32, true>(a, gx, l->F, st)
                                : launch_fused_fwd<32, 32, 32, false>(a, gx, l->F, st);
    case 2: return l->act == 1 ? launch_fused_fwd<32, 32, 8, true>(a, gx, l->F, st)
                                : launch_fused_fwd<32, 32, 8, false>(a, gx, l->F, st);
    case 3: return l->act == 1 ? launch_fused_fwd<64, 64, 16, true>(a, gx, l->F, st)
                                : launch_fused_fwd<64, 64, 16, false>(a, gx, l->F, st);
    default: return DIB_E_UNSUPPORTED;
  }
}

template <int H1, int H2, int E, bool RELU>
static int launch_fused_bwd(const DibFusedBwdArgs& a, int gx, int F, hipStream_t st) {
  using C = DibFusedBwdCfg<H1, H2, E>;
  const size_t lds = (size_t)C::LDS_FLOATS * sizeof(float);
  static std::atomic<bool> attr_set[64];
  if (AttrOnce once(attr_set); once) {
    hipError_t e = hipFuncSetAttribute((const void*)dib_fused_encoder_bwd_kernel<H1, H2, E, RELU>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  DIB_LAUNCH((dib_fused_encoder_bwd_kernel<H1, H2, E, RELU>), dim3(gx, F), dim3(512), lds, st, a);
  return (int)hipGetLastError();
}

// fused backward dgrad chain is instantiated for the configs whose E is a multiple of 32
static bool fused_bwd_ok(const dib_layout* l) {
  if (!(l->fused_id == 0 || l->fused_id == 1)) return false;
  for (int f = 0; f < l->F; ++f)
    if (l->in_dim[f] > 15) return false;  // row in_dim of the 16-row d(W1|b1) tile carries the bias gradient
  return true;
}


static int fused_encoder_bwd(dib_layout* l, const dib_layout::WsMap& m, float* w, int batch, const float* params,
                             const float* beta_dev, float inv_bg, hipStream_t st) {
  DibFusedBwdArgs a;
  a.P = w + m.P; a.batch = batch; a.params = params;
  a.w_off = l->dev_fused_offs; a.b_off = l->dev_fused_offs + 3 * l->F; a.featmap = l->dev_featmap; a.act = l->act;
  a.h2mask = (const unsigned long long*)(w + m.h2mask); a.h1mask = (const unsigned long long*)(w + m.h1mask);
  a.enc_out = w + m.enc_out; a.U = w + m.U; a.GU = w + m.g_u;
  a.dout = w + m.dout; a.dh2 = w + m.g_enc_h[1]; a.dw1_partial = w + m.dw1_partial;
  a.beta_dev = beta_dev; a.inv_bg = inv_bg; a.F = l->F;
  const int gx = fused_gx(l, batch);
  ProfScope ps(kProfFusedBwd, st);
  switch (l->fused_id) {
    case 0: return l->act == 1 ? launch_fused_bwd<128, 128, 32, true>(a, gx, l->F, st)
                                : launch_fused_bwd<128, 128, 32, false>(a, gx, l->F, st);
    case 1: return l->act == 1 ? launch_fused_bwd<32, 32, 32, true>(a, gx, l->F, st)
                                : launch_fused_bwd<32, 32, 32, false>(a, gx, l->F, st);
    default: return DIB_E_UNSUPPORTED;
  }
}

// ---- small-batch row-tile path (dib_small.h) ---------------------------------------------------------------------
// dynamic LDS above 64 KB needs hipFuncAttributeMaxDynamicSharedMemorySize (per device): raised to what a launch needs
static int ensure_dynamic_lds(const void* fn, size_t bytes, int (&have)[64]) {
  if (bytes <= 64 * 1024) return DIB_OK;
  std::lock_guard<std::mutex> lk(g_attr_mu);   // `have` is shared by every host thread (include/dib_hip.h "Threads")
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if ((int)bytes <= have[dev] || bytes <= 64 * 1024) return DIB_OK;
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) return (int)e;
  have[dev] = (int)bytes;
  return DIB_OK;
}
static int small_tiles(int batch) { return cdiv(batch, DIB_SMALL_ROWS); }
// The row-tile regime: while (row tiles x features) - the encoder kernels' workgroup count - is at most "small_wgs" (512: two
// rounds of the 256 CUs).  Measured crossover of the Keras-path training step against the large-batch kernels, F = 2 .. 64 x
// B = 128 .. 2048 (profiles/r05x_small_batch_crossover.txt): row tiles win at <= 512 (0.47 - 0.93 of the large path's time), lose
// from 640 up (1.05 - 2.4 x); a fixed row limit of 1024 had F = 64 at B = 1024 at 1.8 x and left F = 4 at B = 2048 (the chaos
// notebook's loop) on the large path at 1 / 0.8.
static bool small_regime(const dib_layout* l, int batch) {
  return knobs().small_batch && batch <= kSmallMaxBatch &&
         (long long)small_tiles(batch) * l->F <= std::min(knobs().small_wgs, kSmallMaxEncWgs);
}
static bool use_small_enc(const dib_layout* l, int batch) { return l->sb_enc && small_regime(l, batch); }
static bool use_small_int(const dib_layout* l, int batch) { return l->sb_int && small_regime(l, batch); }
// the backward's d(W1|b1) comes as per-workgroup partials (fused backward or small-batch backward): how many
static int enc_dw1_parts(const dib_layout* l, int batch) {
  if (use_small_enc(l, batch)) return small_tiles(batch);
  return fused_bwd_ok(l) ? fused_gx(l, batch) * 8 : 0;
}
// rows of the KL partial table the forward of this (layout, batch) writes
static int enc_kl_rows(const dib_layout* l, const dib_layout::WsMap& m, int batch) {
  if (use_small_enc(l, batch)) return small_tiles(batch);
  return l->fused_id >= 0 ? fused_gx(l, batch) * 8 : m.kl_blocks;
}

static int small_encoder_fwd(dib_layout* l, const dib_layout::WsMap& m, float* w, const float* x, int64_t ldx,
                             const int32_t* row_idx, int64_t row0, int batch, const float* params, uint64_t seed, uint32_t step,
                             int flags, hipStream_t st) {
  DibSmallEncFwdArgs a;
  a.X = x; a.ldx = ldx; a.row_idx = (const int*)row_idx; a.row0 = row0; a.batch = batch; a.params = params;
  a.w_off = l->dev_fused_offs; a.b_off = l->dev_fused_offs + 3 * l->F; a.featmap = l->dev_featmap;
  a.n_blocks = l->n_blocks; a.act = l->act; a.F = l->F; a.E = l->E; a.H1 = l->enc_units[0]; a.H2 = l->enc_units[1];
  const bool infer = (flags & DIB_FWD_INFERENCE) != 0;
  a.P = infer ? nullptr : w + m.P; a.h1 = infer ? nullptr : w + m.enc_h[0]; a.h2 = infer ? nullptr : w + m.enc_h[1];
  a.enc_out = w + m.enc_out; a.U = w + m.U; a.kl_partial = w + m.kl_partial;
  a.seed = seed; a.step = step; a.deterministic = flags & DIB_FWD_DETERMINISTIC; a.step_dev = l->step_dev;
  const size_t lds = (size_t)DIB_SMALL_ROWS * (20 + dib_small_pitch(a.H1) + dib_small_pitch(a.H2) + dib_small_pitch(2 * a.E)) * sizeof(float) +
                     (size_t)DIB_SMALL_XCH_FLOATS_WIDE * sizeof(float);
  static int lds_have[64] = {};
  if (int rc = ensure_dynamic_lds((const void*)dib_small_encoder_fwd_kernel, lds, lds_have)) return rc;
  ProfScope ps(kProfOther, st);
  DIB_LAUNCH(dib_small_encoder_fwd_kernel, dim3(small_tiles(batch), l->F), dim3(DIB_SMALL_THREADS), lds, st, a);
  return (int)hipGetLastError();
}

static int small_encoder_bwd(dib_layout* l, const dib_layout::WsMap& m, float* w, int batch, const float* params,
                             const float* beta_dev, float inv_bg, hipStream_t st) {
  DibSmallEncBwdArgs a;
  a.P = w + m.P; a.batch = batch; a.params = params;
  a.w_off = l->dev_fused_offs; a.b_off = l->dev_fused_offs + 3 * l->F; a.featmap = l->dev_featmap;
  a.act = l->act; a.F = l->F; a.E = l->E; a.H1 = l->enc_units[0]; a.H2 = l->enc_units[1];
  a.h1 = w + m.enc_h[0]; a.h2 = w + m.enc_h[1]; a.enc_out = w + m.enc_out; a.U = w + m.U; a.GU = w + m.g_u;
  a.dout = w + m.dout; a.dh2 = w + m.g_enc_h[1]; a.dw1_partial = w + m.dw1_partial; a.beta_dev = beta_dev; a.inv_bg = inv_bg;
  const size_t lds = (size_t)DIB_SMALL_ROWS * (20 + 2 * dib_small_pitch(a.H1) + 2 * dib_small_pitch(a.H2) + dib_small_pitch(2 * a.E)) * sizeof(float) +
                     (size_t)DIB_SMALL_XCH_FLOATS_WIDE * sizeof(float);
  static int lds_have[64] = {};
  if (int rc = ensure_dynamic_lds((const void*)dib_small_encoder_bwd_kernel, lds, lds_have)) return rc;
  ProfScope ps(kProfOther, st);
  DIB_LAUNCH(dib_small_encoder_bwd_kernel, dim3(small_tiles(batch), l->F), dim3(DIB_SMALL_THREADS), lds, st, a);
  return (int)hipGetLastError();
}

// ---- all weight gradients of a step in ONE grouped launch ------------------------------------------------------------
// Descriptors with absolute workspace offsets for this batch size (the activation buffers' offsets are not linear in the batch:
// every buffer is 256-byte aligned), A and B both relative to the workspace base, C / bias_out relative to the gradient target.
static const std::vector<DibGemmGroup>& wg_table_host(const dib_layout* l, const dib_layout::WsMap& m, int batch) {
  std::lock_guard<std::mutex> lk(l->wg_mu);   // std::map nodes are stable: the reference outlives the lock
  auto it = l->wg_tables.find(batch);
  if (it != l->wg_tables.end()) return it->second;
  std::vector<DibGemmGroup> t;
  const int64_t B = batch;
  for (int ly = 1; ly <= l->n_enc; ++ly) {
    const int win = l->enc_width[ly - 1], wout = l->enc_width[ly];
    for (int f = 0; f < l->F; ++f)
      t.push_back(make_group(fixed_off(m.enc_h[ly - 1] + (int64_t)f * win * B), win,
                             fixed_off((ly == l->n_enc ? m.dout : m.g_enc_h[ly]) + (int64_t)f * wout * B), wout,
                             fixed_off(l->enc_w_off[ly][f]), wout, l->enc_b_off[ly][f], Off(), 0, win, wout, -1));
  }
  for (int ly = 0; ly <= l->n_int; ++ly) {
    const int win = ly == 0 ? l->F * l->E : l->int_width[ly - 1], wout = l->int_width[ly];
    t.push_back(make_group(fixed_off(ly == 0 ? m.U : m.int_h[ly - 1]), win, fixed_off(ly == l->n_int ? m.g_pred : m.g_int_h[ly]), wout,
                           fixed_off(l->int_w_off[ly]), wout, l->int_b_off[ly], Off(), 0, win, wout, -1));
  }
  return l->wg_tables.emplace(batch, std::move(t)).first->second;
}

// groups [first, first + count) of the table (see dib_layout::wg_groups) into the gradient target gt
static int merged_wgrad(dib_layout* l, const dib_layout::WsMap& m, float* w, int batch, float* gt, int first, int count,
                        hipStream_t st) {
  if (count <= 0) return DIB_OK;
  const auto& host = wg_table_host(l, m, batch);
  GemmCall c;
  c.first = 0; c.count = count;
  for (int i = first; i < first + count; ++i) { c.max_m = std::max(c.max_m, host[i].M); c.max_n = std::max(c.max_n, host[i].N); }
  const DibGemmGroup* dev = reinterpret_cast<const DibGemmGroup*>(w + m.wg_table) + first;
  int ns_used = m.nsplit;
  int rc = launch_gemm<2>(dev, c, w, w, gt, nullptr, nullptr, gt, batch, 0, m.nsplit, m.rows_per_split, align_up(l->n_params, 4), st,
                          /*auto_split=*/true, m.nsplit, &ns_used);
  if (rc == DIB_OK && m.nsplit > 1)
    rc = retire_stale_slabs(l, batch, host.data() + first, count, ns_used, m.nsplit, gt, align_up(l->n_params, 4), st);
  return rc;
}
// when one grouped launch for all weight gradients pays: the small-batch regime, where every launch is latency
static bool use_merged_wgrad(const dib_layout* l, int batch) {
  return use_small_enc(l, batch) && use_small_int(l, batch);
}

// A second, independent network for the NEXT dib_small_integration_kernel launch of this thread to carry in its grid
// (dib_integration_fwd_and_mlp_fwd / dib_backward_and_mlp_bwd arm it; whoever armed it launches it alone if nobody took it).
struct SmallCompanion { DibSmallIntArgs args; size_t lds = 0; bool armed = false; };
static thread_local SmallCompanion t_companion;

// workgroups per row tile of a row-tile network launch (dib_small.h "cluster mode"; 1 = the single-workgroup kernel): "int_cluster"
// while the launch stays within "int_cluster_wgs" workgroups, the network's hidden layers hold at least "int_cluster_min_weights"
// weights (below that a layer is a few microseconds on one CU and the exchanges cost more than they save) and the wider exchange
// buffer fits the LDS
static int small_cluster_size(const DibSmallIntArgs& a, size_t lds_bytes, int other_wgs = 0) {
  int cl = std::min(knobs().int_cluster, DIB_SMALL_CL_MAX);
  const int budget = std::min(knobs().int_cluster_wgs, device_cus());   // one workgroup per CU (141 KB of LDS each)
  // more row tiles: 4 per tile instead of 8 while the launch (with the other network of a paired grid: other_wgs) stays within the
  // budget - every workgroup must be resident for the networks to run side by side; 2 per tile measured no gain
  // (profiles/r06u_int_cluster_sweep.txt)
  while (cl > 4 && small_tiles(a.batch) * cl + other_wgs > budget) cl >>= 1;
  if (cl <= 1 || small_tiles(a.batch) * cl + other_wgs > budget || lds_bytes > 160 * 1024) return 1;
  if (a.mode & (DIB_SMALL_INT_HEAD_REDUCE)) return 1;   // (its last-arriver reduce counts workgroups, not tiles)
  long long weights = 0;
  for (int i = 0, k = a.K0; i < a.n_hidden; k = a.width[i], ++i) weights += (long long)k * a.width[i];
  return weights >= knobs().int_cluster_min_weights ? cl : 1;
}

// one launch of dib_small_integration_kernel; `mode` = DIB_SMALL_INT_* bits.  Head arguments may be null / 0 without a head.
static int small_integration(dib_layout* l, const dib_layout::WsMap& m, float* w, int batch, const float* params, int mode,
                             int loss_kind, const float* y, int64_t ldy, const int32_t* row_idx, int64_t row0, float inv_bg,
                             hipStream_t st) {
  DibSmallIntArgs a;
  std::memset(&a, 0, sizeof(a));
  a.U = w + m.U; a.GU = w + m.g_u; a.batch = batch; a.K0 = l->F * l->E; a.params = params;
  a.n_hidden = l->n_int;
  for (int i = 0; i < l->n_int; ++i) { a.width[i] = l->int_units[i]; a.h[i] = w + m.int_h[i]; a.g[i] = w + m.g_int_h[i]; }
  for (int i = 0; i <= l->n_int; ++i) { a.w_off[i] = l->int_w_off[i]; a.b_off[i] = l->int_b_off[i]; }
  a.width[l->n_int] = l->out_dim;
  a.act = l->act; a.out_act = l->out_act; a.out_dim = l->out_dim; a.mode = mode;
  a.pred = w + m.pred; a.g_pred = w + m.g_pred;
  a.loss_kind = loss_kind; a.Y = y; a.ldy = ldy; a.row_idx = (const int*)row_idx; a.row0 = row0; a.inv_bg = inv_bg;
  a.partial_w = w + m.skinny_partial; a.partial_l = w + m.loss_partial;
  // cluster mode: few row tiles, each on `cl` workgroups (dib_small.h)
  const size_t cl_extra = (size_t)(DIB_SMALL_XCH_FLOATS_WIDE - DIB_SMALL_XCH_FLOATS) * sizeof(float);
  int cl = small_cluster_size(a, (size_t)l->sb_int_lds + cl_extra);
  if (t_companion.armed) {
    t_companion.armed = false;
    DibSmallIntPair p;
    p.s[0] = a; p.s[1] = t_companion.args;
    // the companion clusters by the same rule on its own size (training launches only: it has no exchange buffers for a launch
    // without stashes); its arrival counters are the second half of this workspace's
    DibSmallIntArgs& c = p.s[1];
    int ccl = (c.mode & DIB_SMALL_INT_INFER) || small_tiles(c.batch) > small_tiles(batch) ? 1 : small_cluster_size(c, t_companion.lds + cl_extra);
    // the two networks run side by side only while all their workgroups are resident (one per CU): the companion first gives up
    // its cluster, then this network sizes itself next to it
    if (small_tiles(batch) * cl + small_tiles(c.batch) * ccl > std::min(knobs().int_cluster_wgs, device_cus())) {
      ccl = 1;
      cl = small_cluster_size(a, (size_t)l->sb_int_lds + cl_extra, small_tiles(c.batch));
    }
    if (cl > 1 || ccl > 1) {
      p.s[0].cl = cl; p.s[0].cl_sync = (unsigned*)(w + m.cl_sync);
      p.s[0].cl_agent_scope = c.cl_agent_scope = knobs().int_cluster_short_exchange ? 0 : 1;
      for (int i = 0; i < l->n_int; ++i) p.s[0].xh[i] = w + m.cl_x[i];
      c.cl = ccl; c.cl_sync = (unsigned*)(w + m.cl_sync) + (size_t)small_tiles(batch) * DIB_SMALL_CL_SYNC_WORDS;
      const size_t lds = std::max((size_t)l->sb_int_lds, t_companion.lds) + cl_extra;
      static int pc_lds_have[64] = {};
      if (int rc = ensure_dynamic_lds((const void*)dib_small_integration_pair_cluster_kernel, lds, pc_lds_have)) return rc;
      ProfScope ps(kProfOther, st);
      const int gx = std::max(8 * cdiv(small_tiles(batch), 8) * cl, 8 * cdiv(small_tiles(c.batch), 8) * ccl);
      DIB_LAUNCH(dib_small_integration_pair_cluster_kernel, dim3(gx, 2), dim3(DIB_SMALL_THREADS), lds, st, p);
      return (int)hipGetLastError();
    }
    const size_t lds = std::max((size_t)l->sb_int_lds, t_companion.lds);
    static int pair_lds_have[64] = {};
    if (int rc = ensure_dynamic_lds((const void*)dib_small_integration_pair_kernel, lds, pair_lds_have)) return rc;
    ProfScope ps(kProfOther, st);
    DIB_LAUNCH(dib_small_integration_pair_kernel, dim3(std::max(small_tiles(batch), small_tiles(p.s[1].batch)), 2),
               dim3(DIB_SMALL_THREADS), lds, st, p);
    return (int)hipGetLastError();
  }
  if (cl > 1) {
    a.cl = cl; a.cl_sync = (unsigned*)(w + m.cl_sync); a.cl_agent_scope = knobs().int_cluster_short_exchange ? 0 : 1;
    for (int i = 0; i < l->n_int; ++i) a.xh[i] = w + m.cl_x[i];
    const size_t cl_lds = (size_t)l->sb_int_lds + cl_extra;
    static int cl_lds_have[64] = {};
    if (int rc = ensure_dynamic_lds((const void*)dib_small_integration_cluster_kernel, cl_lds, cl_lds_have)) return rc;
    ProfScope ps(kProfOther, st);
    DIB_LAUNCH(dib_small_integration_cluster_kernel, dim3(8 * cdiv(small_tiles(batch), 8) * cl), dim3(DIB_SMALL_THREADS), cl_lds, st, a);
    return (int)hipGetLastError();
  }
  static int lds_have[64] = {};
  if (int rc = ensure_dynamic_lds((const void*)dib_small_integration_kernel, (size_t)l->sb_int_lds, lds_have)) return rc;
  ProfScope ps(kProfOther, st);
  DIB_LAUNCH(dib_small_integration_kernel, dim3(small_tiles(batch)), dim3(DIB_SMALL_THREADS), (size_t)l->sb_int_lds, st, a);
  return (int)hipGetLastError();
}

extern "C" {

const char* dib_version(void) { return kVersion; }
int dib_abi_version(void) { return DIB_ABI_VERSION; }

const char* dib_error_string(int code) {
  switch (code) {
    case DIB_OK: return "ok";
    case DIB_E_ARG: return "invalid argument";
    case DIB_E_SHAPE: return "shape mismatch";
    case DIB_E_WORKSPACE: return "workspace / descriptor tables missing";
    case DIB_E_UNSUPPORTED: return "unsupported configuration";
    case DIB_E_NODEVICE: return "no HIP device";
    default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown dib error";
  }
}

int dib_layout_create(int F, const int* feature_dims, int n_enc, const int* enc_units, int E, int n_int,
                      const int* int_units, int out_dim, int use_pe, int n_freq, int act, int out_act,
                      dib_layout** out) {
  if (!out || F <= 0 || !feature_dims || n_enc < 0 || n_int < 0 || E <= 0 || out_dim <= 0) return DIB_E_ARG;
  if ((n_enc > 0 && !enc_units) || (n_int > 0 && !int_units)) return DIB_E_ARG;
  if (!act_ok(act) || !act_ok(out_act)) return DIB_E_UNSUPPORTED;
  if ((E + 3) / 4 > 256) return DIB_E_UNSUPPORTED;
  dib_layout* l = new (std::nothrow) dib_layout();
  if (!l) return DIB_E_ARG;
  l->F = F; l->n_enc = n_enc; l->E = E; l->n_int = n_int; l->out_dim = out_dim;
  l->use_pe = use_pe ? 1 : 0; l->n_freq = n_freq; l->act = act; l->out_act = out_act;
  l->dims.assign(feature_dims, feature_dims + F);
  l->enc_units.assign(enc_units, enc_units + n_enc);
  l->int_units.assign(int_units, int_units + n_int);
  // reference models.py:70: frequencies = 2**arange(1, n_freq) -> n_freq-1 sinusoids
  l->n_blocks = (l->use_pe && n_freq > 1) ? n_freq : 1;
  for (int f = 0; f < F; ++f) {
    if (l->dims[f] <= 0) { delete l; return DIB_E_ARG; }
    l->x_off.push_back(l->sum_d);
    l->in_off.push_back(l->pw);
    l->in_dim.push_back(l->dims[f] * l->n_blocks);
    l->sum_d += l->dims[f];
    l->pw += l->dims[f] * l->n_blocks;
    for (int c = 0; c < l->dims[f]; ++c) l->colmap.push_back(make_int4(f, c, l->dims[f], l->in_off[f]));
  }
  for (int i = 0; i < n_enc; ++i) { if (enc_units[i] <= 0) { delete l; return DIB_E_ARG; } l->enc_width.push_back(enc_units[i]); }
  l->enc_width.push_back(2 * E);
  for (int i = 0; i < n_int; ++i) { if (int_units[i] <= 0) { delete l; return DIB_E_ARG; } l->int_width.push_back(int_units[i]); }
  l->int_width.push_back(out_dim);

  // ---- flat parameter layout: per encoder layer {all kernels (feature-major), all biases}, then integration ----
  int64_t o = 0;
  const int LE = n_enc + 1, LI = n_int + 1;
  l->enc_w_off.assign(LE, std::vector<int64_t>(F));
  l->enc_b_off.assign(LE, std::vector<int64_t>(F));
  for (int ly = 0; ly < LE; ++ly) {
    const int wout = l->enc_width[ly];
    for (int f = 0; f < F; ++f) {
      const int win = ly == 0 ? l->in_dim[f] : l->enc_width[ly - 1];
      o = align_up(o, 4);
      l->enc_w_off[ly][f] = o;
      o += (int64_t)win * wout;
    }
    o = align_up(o, 4);
    for (int f = 0; f < F; ++f) { l->enc_b_off[ly][f] = o; o += wout; }
  }
  for (int ly = 0; ly < LI; ++ly) {
    const int win = ly == 0 ? F * E : l->int_width[ly - 1];
    const int wout = l->int_width[ly];
    o = align_up(o, 4);
    l->int_w_off.push_back(o);
    o += (int64_t)win * wout;
    o = align_up(o, 4);
    l->int_b_off.push_back(o);
    o += wout;
  }
  l->n_params = o;

  // ---- GEMM group descriptors.  Encoder-bank activations are FEATURE-MAJOR: [F][B][width], i.e. feature f's
  // operand is the dense matrix at element offset (f*width)*B (ragged first layer: in_off[f]*B).  U / g_u (the
  // integration network's operand, reference models.py:122 tf.concat) stay sample-major [B, F*E]. ----
  auto& T = l->table;
  for (int ly = 0; ly < LE; ++ly) {
    const int wout = l->enc_width[ly];
    GemmCall fw, dg, wg;
    fw.first = (int)T.size();
    for (int f = 0; f < F; ++f) {
      const int win = ly == 0 ? l->in_dim[f] : l->enc_width[ly - 1];
      const Off a = batch_off(ly == 0 ? (int64_t)l->in_off[f] : (int64_t)f * win);
      T.push_back(make_group(a, win, fixed_off(l->enc_w_off[ly][f]), wout, batch_off((int64_t)f * wout), wout,
                             l->enc_b_off[ly][f], Off(), 0, -1, wout, win));
    }
    fw.count = F; fw.max_m = -1; fw.max_n = wout;
    l->enc_fwd.push_back(fw);
    // dgrad (ly >= 1): g_in[B, win] = (g_out[B, wout] @ W[win, wout]^T) * act'(h_in)
    dg.first = (int)T.size();
    if (ly >= 1) {
      const int win = l->enc_width[ly - 1];
      for (int f = 0; f < F; ++f)
        T.push_back(make_group(batch_off((int64_t)f * wout), wout, fixed_off(l->enc_w_off[ly][f]), wout,
                               batch_off((int64_t)f * win), win, -1, batch_off((int64_t)f * win), win, -1, win, wout));
      dg.count = F; dg.max_m = -1; dg.max_n = win;
    }
    l->enc_dgrad.push_back(dg);
    // wgrad: dW[win, wout] = h_in[B, win]^T @ g_out[B, wout] ; db = colsum(g_out)
    wg.first = (int)T.size();
    int max_in = 0;
    for (int f = 0; f < F; ++f) {
      const int win = ly == 0 ? l->in_dim[f] : l->enc_width[ly - 1];
      const Off a = batch_off(ly == 0 ? (int64_t)l->in_off[f] : (int64_t)f * win);
      T.push_back(make_group(a, win, batch_off((int64_t)f * wout), wout, fixed_off(l->enc_w_off[ly][f]), wout,
                             l->enc_b_off[ly][f], Off(), 0, win, wout, -1));
      max_in = std::max(max_in, win);
    }
    wg.count = F; wg.max_m = max_in; wg.max_n = wout;
    l->max_wgrad_tiles64 = std::max(l->max_wgrad_tiles64, (long long)cdiv(max_in, 64) * cdiv(wout, 64) * F);
    l->enc_wgrad.push_back(wg);
  }
  for (int ly = 0; ly < LI; ++ly) {
    const int win = ly == 0 ? F * E : l->int_width[ly - 1];
    const int wout = l->int_width[ly];
    GemmCall fw, dg, wg;
    fw.first = (int)T.size();
    T.push_back(make_group(Off(), win, fixed_off(l->int_w_off[ly]), wout, Off(), wout, l->int_b_off[ly], Off(), 0, -1,
                           wout, win));
    fw.count = 1; fw.max_m = -1; fw.max_n = wout;
    l->int_fwd.push_back(fw);
    dg.first = (int)T.size();
    T.push_back(make_group(Off(), wout, fixed_off(l->int_w_off[ly]), wout, Off(), win, -1, Off(), win, -1, win, wout));
    dg.count = 1; dg.max_m = -1; dg.max_n = win;
    l->int_dgrad.push_back(dg);
    wg.first = (int)T.size();
    T.push_back(make_group(Off(), win, Off(), wout, fixed_off(l->int_w_off[ly]), wout, l->int_b_off[ly], Off(), 0, win,
                           wout, -1));
    wg.count = 1; wg.max_m = win; wg.max_n = wout;
    l->max_wgrad_tiles64 = std::max(l->max_wgrad_tiles64, (long long)cdiv(win, 64) * cdiv(wout, 64));
    l->int_wgrad.push_back(wg);
  }
  // fused encoder-bank path: two hidden layers, instantiated (H1,H2,E), encoder inputs <= 16 wide
  {
    static const int kFused[][3] = {{128, 128, 32}, {32, 32, 32}, {32, 32, 8}, {64, 64, 16}};
    bool in_ok = true;
    for (int f = 0; f < F; ++f) in_ok = in_ok && l->in_dim[f] <= 16;
    if (knobs().fused_encoder && n_enc == 2 && in_ok && act >= 0 && act <= 2)
      for (int i = 0; i < 4; ++i)
        if (kFused[i][0] == enc_units[0] && kFused[i][1] == enc_units[1] && kFused[i][2] == E) l->fused_id = i;
    for (int ly = 0; ly < LE && ly < 3; ++ly)
      for (int f = 0; f < F; ++f) l->fused_offs.push_back(l->enc_w_off[ly][f]);
    for (int ly = LE; ly < 3; ++ly)
      for (int f = 0; f < F; ++f) l->fused_offs.push_back(0);
    for (int ly = 0; ly < LE && ly < 3; ++ly)
      for (int f = 0; f < F; ++f) l->fused_offs.push_back(l->enc_b_off[ly][f]);
    for (int ly = LE; ly < 3; ++ly)
      for (int f = 0; f < F; ++f) l->fused_offs.push_back(0);
    for (int f = 0; f < F; ++f) l->featmap.push_back(make_int4(l->dims[f], l->in_dim[f], l->x_off[f], l->in_off[f]));
  }
  // small-batch row-tile kernels (dib_small.h): two-hidden-layer encoders of widths % 16 == 0 with inputs <= 15 wide (the 16th
  // row of the d(W1|b1) tile carries the bias gradient), linear / relu / leaky_relu; integration networks of 1-3 hidden layers of widths
  // % 16 == 0 (<= 1024 for the head's lane-strided dot) whose 16-row activation tiles fit the CU's LDS
  {
    bool in_ok = true;
    for (int f = 0; f < F; ++f) in_ok = in_ok && l->in_dim[f] <= 15;
    const bool pl_act = act >= 0 && act <= 2 && out_act >= 0 && out_act <= 2;   // piecewise-linear activations (dib_small.h)
    l->sb_enc = pl_act && n_enc == 2 && in_ok && enc_units[0] % 16 == 0 && enc_units[1] % 16 == 0 && (2 * E) % 16 == 0 &&
                enc_units[0] <= 1024 && enc_units[1] <= 1024 && E <= 512;
    bool w_ok = n_int >= 1 && n_int <= 3 && (F * E) % 16 == 0;
    int64_t fl = (int64_t)DIB_SMALL_ROWS * dib_small_pitch(F * E);
    for (int i = 0; i < n_int && w_ok; ++i) {
      w_ok = int_units[i] % 16 == 0 && int_units[i] <= 1024;
      fl += 2ll * DIB_SMALL_ROWS * dib_small_pitch(int_units[i]);
    }
    fl += (int64_t)DIB_SMALL_ROWS * dib_small_pitch(out_dim) + DIB_SMALL_XCH_FLOATS + (w_ok ? 9 * (int_units[n_int - 1] + 1) + 32 : 0);
    l->sb_int = pl_act && w_ok && fl * 4 <= 150 * 1024;
    l->sb_int_lds = (int)(fl * 4);
  }
  *out = l;
  return DIB_OK;
}

void dib_layout_destroy(dib_layout* l) {
  if (!l) return;
  delete l;
}

int64_t dib_layout_param_count(const dib_layout* l) { return l ? l->n_params : DIB_E_ARG; }

int dib_layout_param_block(const dib_layout* l, int net, int layer, int feature, int what, int64_t* offset,
                           int* rows, int* cols) {
  if (!l || !offset || !rows || !cols) return DIB_E_ARG;
  if (net == 0) {
    if (layer < 0 || layer > l->n_enc || feature < 0 || feature >= l->F) return DIB_E_ARG;
    const int win = layer == 0 ? l->in_dim[feature] : l->enc_width[layer - 1];
    const int wout = l->enc_width[layer];
    if (what == 0) { *offset = l->enc_w_off[layer][feature]; *rows = win; *cols = wout; }
    else { *offset = l->enc_b_off[layer][feature]; *rows = 1; *cols = wout; }
    return DIB_OK;
  }
  if (net == 1) {
    if (layer < 0 || layer > l->n_int) return DIB_E_ARG;
    const int win = layer == 0 ? l->F * l->E : l->int_width[layer - 1];
    const int wout = l->int_width[layer];
    if (what == 0) { *offset = l->int_w_off[layer]; *rows = win; *cols = wout; }
    else { *offset = l->int_b_off[layer]; *rows = 1; *cols = wout; }
    return DIB_OK;
  }
  return DIB_E_ARG;
}

int64_t dib_layout_table_bytes(const dib_layout* l) {
  if (!l) return DIB_E_ARG;
  return align_up((int64_t)l->table.size() * sizeof(DibGemmGroup), 256) +
         align_up((int64_t)l->colmap.size() * sizeof(int4), 256) +
         align_up((int64_t)l->fused_offs.size() * sizeof(long long), 256) +
         align_up((int64_t)l->featmap.size() * sizeof(int4), 256);
}

int dib_layout_upload_tables(dib_layout* l, void* dev_tables, dib_stream_t stream) {
  if (!l || !dev_tables) return DIB_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int64_t gbytes = (int64_t)l->table.size() * sizeof(DibGemmGroup);
  char* base = (char*)dev_tables;
  hipError_t e = hipMemcpyAsync(base, l->table.data(), gbytes, hipMemcpyHostToDevice, st);
  if (e != hipSuccess) return (int)e;
  char* cm = base + align_up(gbytes, 256);
  e = hipMemcpyAsync(cm, l->colmap.data(), l->colmap.size() * sizeof(int4), hipMemcpyHostToDevice, st);
  if (e != hipSuccess) return (int)e;
  char* fo = cm + align_up((int64_t)l->colmap.size() * sizeof(int4), 256);
  e = hipMemcpyAsync(fo, l->fused_offs.data(), l->fused_offs.size() * sizeof(long long), hipMemcpyHostToDevice, st);
  if (e != hipSuccess) return (int)e;
  char* fmp = fo + align_up((int64_t)l->fused_offs.size() * sizeof(long long), 256);
  e = hipMemcpyAsync(fmp, l->featmap.data(), l->featmap.size() * sizeof(int4), hipMemcpyHostToDevice, st);
  if (e != hipSuccess) return (int)e;
  l->dev_groups = (const DibGemmGroup*)base;
  l->dev_colmap = (const int4*)cm;
  l->dev_fused_offs = (const long long*)fo;
  l->dev_featmap = (const int4*)fmp;
  return DIB_OK;
}

int dib_layout_set_step_counter(dib_layout* l, const uint32_t* step_dev) {
  if (!l) return DIB_E_ARG;
  l->step_dev = (const unsigned*)step_dev;
  return DIB_OK;
}

int64_t dib_workspace_bytes(const dib_layout* l, int batch) {
  if (!l || batch <= 0) return DIB_E_ARG;
  return l->map(batch).total * (int64_t)sizeof(float);
}

int dib_workspace_init(const dib_layout* l, int batch, void* ws, dib_stream_t stream) {
  if (!l || !ws || batch <= 0) return DIB_E_ARG;
  const auto m = l->map(batch);
  // the arrival counters of dib_step_tail (self-cleaning afterwards)
  hipError_t e0 = hipMemsetAsync((float*)ws + m.sync, 0, (size_t)DIB_TAIL_SYNC_WORDS * sizeof(unsigned), (hipStream_t)stream);
  if (e0 != hipSuccess) return (int)e0;
  if (l->sb_int && batch <= kSmallMaxBatch) {   // ... and of the integration kernel's cluster mode
    e0 = hipMemsetAsync((float*)ws + m.cl_sync, 0, 2 * (size_t)cdiv(batch, DIB_SMALL_ROWS) * DIB_SMALL_CL_SYNC_WORDS * sizeof(unsigned),
                        (hipStream_t)stream);
    if (e0 != hipSuccess) return (int)e0;
  }
  {  // the merged weight-gradient table of this batch size (the host copy lives in the layout: the copy may be asynchronous)
    const auto& t = wg_table_host(l, m, batch);
    e0 = hipMemcpyAsync((float*)ws + m.wg_table, t.data(), t.size() * sizeof(DibGemmGroup), hipMemcpyHostToDevice, (hipStream_t)stream);
    if (e0 != hipSuccess) return (int)e0;
  }
  // the per-step scalars: a caller that accumulates only the KL terms (custom loss) must not pick up stale loss sums
  e0 = hipMemsetAsync((float*)ws + m.step_out, 0, (size_t)(l->F + 3) * sizeof(float), (hipStream_t)stream);
  if (e0 != hipSuccess) return (int)e0;
  if (m.nsplit <= 1) return DIB_OK;
  // the split-batch weight-gradient slabs: dib_grads_finalize sums all nsplit slabs of every block, including the slabs
  // a launch never writes (halved splits of narrow layers, slabs >= 1 of the skinny output layer, the layer-1 block
  // under the fused backward, alignment gaps) - those must read as zero.  Nothing ever writes a non-zero there.
  return (int)hipMemsetAsync((float*)ws + m.wgrad_partial, 0,
                             (size_t)m.nsplit * (size_t)align_up(l->n_params, 4) * sizeof(float), (hipStream_t)stream);
}

int64_t dib_workspace_offset(const dib_layout* l, int batch, int which) {
  if (!l || batch <= 0) return DIB_E_ARG;
  const auto m = l->map(batch);
  int64_t o = -1;
  switch (which) {
    case DIB_WS_U: o = m.U; break;
    case DIB_WS_PRED: o = m.pred; break;
    case DIB_WS_ENC_OUT: o = m.enc_out; break;
    case DIB_WS_G_U: o = m.g_u; break;
    case DIB_WS_STEP_OUT: o = m.step_out; break;
    case DIB_WS_G_PRED: o = m.g_pred; break;
    default:
      if (which >= DIB_WS_ENC_H0 && which < DIB_WS_ENC_H0 + l->n_enc) o = m.enc_h[which - DIB_WS_ENC_H0];
      else if (which >= DIB_WS_INT_H0 && which < DIB_WS_INT_H0 + l->n_int) o = m.int_h[which - DIB_WS_INT_H0];
      else return DIB_E_ARG;
  }
  return o * (int64_t)sizeof(float);
}

int dib_layout_wgrad_splits(const dib_layout* l, int batch) {
  if (!l || batch <= 0) return DIB_E_ARG;
  return l->map(batch).nsplit;
}

// ---- forward ---------------------------------------------------------------------------------
static int encoder_chain_fwd(dib_layout* l, const dib_layout::WsMap& m, float* w, int batch, const float* params,
                             int first_group_offset, int group_count, hipStream_t st) {
  const int LE = l->n_enc + 1;
  for (int ly = 0; ly < LE; ++ly) {
    GemmCall c = l->enc_fwd[ly];
    c.first += first_group_offset;
    c.count = group_count;
    const float* A = ly == 0 ? w + m.P : w + m.enc_h[ly - 1];
    float* C = ly == LE - 1 ? w + m.enc_out : w + m.enc_h[ly];
    const int act = ly == LE - 1 ? DIB_ACT_LINEAR : l->act;  // reference models.py:78: last Dense(2E) is linear
    int rc = launch_gemm<0>(l, c, A, params, C, params, nullptr, nullptr, batch, act, 1, 0, 0, st);
    if (rc) return rc;
  }
  return DIB_OK;
}

int dib_encoder_bank_fwd(dib_layout* l, const float* x, int64_t ldx, const int32_t* row_idx, int64_t row0, int batch,
                         const float* params, uint64_t seed, uint32_t step, int deterministic, void* ws,
                         dib_stream_t stream) {
  if (!l || !x || !params || !ws || batch <= 0) return DIB_E_ARG;
  if (!l->dev_groups) return DIB_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const auto m = l->map(batch);
  float* w = (float*)ws;
  if (use_small_enc(l, batch)) {   // gather + positional encoding + Dense chain + reparameterisation + KL partials: one launch
    int rc = small_encoder_fwd(l, m, w, x, ldx, row_idx, row0, batch, params, seed, step, deterministic, st);
    if (rc || (deterministic & DIB_FWD_DEFER_SUMS)) return rc;
    ProfScope ps(kProfOther, st);
    DIB_LAUNCH(dib_colsum_partials_kernel, dim3(l->F), dim3(256), 0, st, w + m.kl_partial, small_tiles(batch), l->F,
                       w + m.step_out);
    return (int)hipGetLastError();
  }
  { ProfScope ps(kProfOther, (hipStream_t)stream);
  if ((long long)cdiv(l->sum_d, 64) * cdiv(batch, 64) >= 512)
    DIB_LAUNCH(dib_posenc_kernel<64>, dim3(cdiv(l->sum_d, 64), cdiv(batch, 64)), dim3(256), 0, st, x, (long long)ldx,
                       (const int*)row_idx, (long long)row0, batch, l->dev_colmap, l->sum_d, l->n_blocks, w + m.P);
  else
    DIB_LAUNCH(dib_posenc_kernel<16>, dim3(cdiv(l->sum_d, 64), cdiv(batch, 16)), dim3(256), 0, st, x, (long long)ldx,
                       (const int*)row_idx, (long long)row0, batch, l->dev_colmap, l->sum_d, l->n_blocks, w + m.P); }
  int rc = (int)hipGetLastError();
  if (rc) return rc;
  if (l->fused_id >= 0) {  // one launch: positional encoding + 3 layers + reparameterisation + KL
    int gx = 1;
    rc = fused_encoder_fwd(l, m, w, x, ldx, row_idx, row0, batch, params, seed, step, deterministic, st, &gx);
    if (rc) return rc;
    if (deterministic & DIB_FWD_DEFER_SUMS) return DIB_OK;   // dib_step_tail(DIB_TAIL_KL) sums the partials
    { ProfScope ps(kProfOther, (hipStream_t)stream);
    DIB_LAUNCH(dib_colsum_partials_kernel, dim3(l->F), dim3(256), 0, st, w + m.kl_partial, gx * 8, l->F,
                       w + m.step_out); }
    return (int)hipGetLastError();
  }
  rc = encoder_chain_fwd(l, m, w, batch, params, 0, l->F, st);
  if (rc) return rc;
  { ProfScope ps(kProfOther, (hipStream_t)stream);
  DIB_LAUNCH(dib_reparam_kl_fwd_kernel, dim3(m.kl_blocks, l->F), dim3(256), 0, st, w + m.enc_out, w + m.U,
                     w + m.kl_partial, (const int*)row_idx, (long long)row0, batch, l->F, l->E,
                     (unsigned long long)seed, (unsigned)step, deterministic & DIB_FWD_DETERMINISTIC, l->step_dev); }
  rc = (int)hipGetLastError();
  if (rc) return rc;
  if (deterministic & DIB_FWD_DEFER_SUMS) return DIB_OK;
  { ProfScope ps(kProfOther, (hipStream_t)stream);
  DIB_LAUNCH(dib_colsum_partials_kernel, dim3(l->F), dim3(256), 0, st, w + m.kl_partial, m.kl_blocks, l->F,
                     w + m.step_out); }
  return (int)hipGetLastError();
}

static int integration_fwd_impl(dib_layout* l, int batch, const float* params, void* ws, dib_stream_t stream,
                                bool with_output_layer) {
  if (!l || !params || !ws || batch <= 0) return DIB_E_ARG;
  if (!l->dev_groups) return DIB_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const auto m = l->map(batch);
  float* w = (float*)ws;
  const int LI = l->n_int + 1;
  int first = 0;
  if (use_small_int(l, batch)) {   // the hidden layers (and a general output layer of width % 16 == 0) in one launch
    const bool out_too = with_output_layer && l->out_dim % 16 == 0;
    int rc = small_integration(l, m, w, batch, params, DIB_SMALL_INT_FWD | (out_too ? DIB_SMALL_INT_OUT : 0), 0, nullptr, 0,
                               nullptr, 0, 0.f, st);
    if (rc || out_too || !with_output_layer) return rc;
    first = LI - 1;   // the narrow output layer below
  }
  for (int ly = first; ly < LI; ++ly) {
    if (ly == LI - 1 && !with_output_layer) break;
    const float* A = ly == 0 ? w + m.U : w + m.int_h[ly - 1];
    float* C = ly == LI - 1 ? w + m.pred : w + m.int_h[ly];
    const int act = ly == LI - 1 ? l->out_act : l->act;  // reference models.py:82-83
    int rc;
    if (ly == LI - 1 && l->out_dim <= DIB_SKINNY_MAX) {  // 1-unit logit & co: HBM-bound stream, not an MFMA tile
      const int win = ly == 0 ? l->F * l->E : l->int_width[ly - 1];
      ProfScope ps(kProfOther, st);
      DIB_LAUNCH(dib_skinny_fwd_kernel, dim3(grid_for((int64_t)batch * 64, 256, 2048)), dim3(256), 0, st, A, batch,
                         win, params + l->int_w_off[ly], params + l->int_b_off[ly], l->out_dim, act, C);
      rc = (int)hipGetLastError();
    } else {
      rc = launch_gemm<0>(l, l->int_fwd[ly], A, params, C, params, nullptr, nullptr, batch, act, 1, 0, 0, st);
    }
    if (rc) return rc;
  }
  return DIB_OK;
}

int dib_integration_fwd(dib_layout* l, int batch, const float* params, void* ws, dib_stream_t stream) {
  return integration_fwd_impl(l, batch, params, ws, stream, true);
}

int dib_integration_fwd_hidden(dib_layout* l, int batch, const float* params, void* ws, dib_stream_t stream) {
  return integration_fwd_impl(l, batch, params, ws, stream, false);
}

// ---- loss + backward ----------------------------------------------------------------------------
int dib_loss_fwd_bwd(dib_layout* l, int loss_kind, const float* y, int64_t ldy, const int32_t* row_idx, int64_t row0,
                     int batch, float inv_global_batch, int flags, void* ws, dib_stream_t stream) {
  if (!l || !y || !ws || batch <= 0) return DIB_E_ARG;
  if (loss_kind < 0 || loss_kind > 3) return DIB_E_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const auto m = l->map(batch);
  float* w = (float*)ws;
  { ProfScope ps(kProfOther, (hipStream_t)stream);
  DIB_LAUNCH(dib_loss_kernel, dim3(m.loss_blocks), dim3(256), 0, st, loss_kind, w + m.pred, l->out_dim, y,
                     (long long)ldy, (const int*)row_idx, (long long)row0, batch, inv_global_batch, l->out_act,
                     w + m.g_pred, w + m.loss_partial); }
  int rc = (int)hipGetLastError();
  if (rc) return rc;
  if (flags & DIB_HEAD_DEFER_SUMS) return DIB_OK;   // dib_step_tail(DIB_TAIL_LOSS) sums the partials
  { ProfScope ps(kProfOther, (hipStream_t)stream);
  DIB_LAUNCH(dib_loss_finalize_kernel, dim3(2), dim3(256), 0, st, (const float*)(w + m.loss_partial), m.loss_blocks,
                     (float)batch, w + m.step_out + l->F); }
  return (int)hipGetLastError();
}

static inline float* wgrad_target(const dib_layout::WsMap& m, float* w, float* grads) {
  return m.nsplit > 1 ? w + m.wgrad_partial : grads;
}

static int integration_bwd_impl(dib_layout* l, int batch, const float* params, float* grads, void* ws, dib_stream_t stream,
                                bool with_output_layer, bool skip_dgrad = false) {
  if (!l || !params || !grads || !ws || batch <= 0) return DIB_E_ARG;
  if (!l->dev_groups) return DIB_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const auto m = l->map(batch);
  float* w = (float*)ws;
  float* gt = wgrad_target(m, w, grads);
  const long long sstride = align_up(l->n_params, 4);
  const int LI = l->n_int + 1;
  // small batches: the whole dgrad chain dL/dpred (or the head's dL/dh) -> dL/du in one launch; the weight gradients below
  if (!skip_dgrad && use_small_int(l, batch) && (!with_output_layer || l->out_dim % 16 == 0)) {
    int rc = small_integration(l, m, w, batch, params, DIB_SMALL_INT_LOAD_H | DIB_SMALL_INT_BWD |
                               (with_output_layer ? DIB_SMALL_INT_BWD_OUT : DIB_SMALL_INT_LOAD_G), 0, nullptr, 0, nullptr, 0, 0.f, st);
    if (rc) return rc;
    skip_dgrad = true;
  }
  for (int ly = LI - 1; ly >= 0; --ly) {
    if (ly == LI - 1 && !with_output_layer) continue;  // done by dib_output_head_fused
    const float* gout = ly == LI - 1 ? w + m.g_pred : w + m.g_int_h[ly];
    const float* hin = ly == 0 ? w + m.U : w + m.int_h[ly - 1];
    float* gin = ly == 0 ? w + m.g_u : w + m.g_int_h[ly - 1];
    int rc;
    if (ly == LI - 1 && l->out_dim <= DIB_SKINNY_MAX) {
      const int win = ly == 0 ? l->F * l->E : l->int_width[ly - 1];
      ProfScope ps(kProfOther, st);
      // stage 1 per row chunk, stage 2 into slab 0 (the other slabs of this block stay zero), both fixed-order
      DIB_LAUNCH(dib_skinny_wgrad_kernel, dim3(m.skinny_chunks), dim3(256), 0, st, hin, gout, batch, win, l->out_dim,
                         m.skinny_rows, w + m.skinny_partial);
      DIB_LAUNCH(dib_skinny_wgrad_reduce_kernel, dim3(win * l->out_dim + l->out_dim), dim3(256),
                         0, st, (const float*)(w + m.skinny_partial), m.skinny_chunks, win, l->out_dim,
                         gt + l->int_w_off[ly], gt + l->int_b_off[ly]);
      DIB_LAUNCH(dib_skinny_dgrad_kernel, dim3(grid_for((int64_t)batch * win)), dim3(256), 0, st, gout, batch, win,
                         params + l->int_w_off[ly], l->out_dim, ly == 0 ? (const float*)nullptr : hin, ly == 0 ? 0 : l->act,
                         gin);
      rc = (int)hipGetLastError();
      if (rc) return rc;
      continue;
    }
    rc = launch_gemm<2>(l, l->int_wgrad[ly], hin, gout, gt, nullptr, nullptr, gt, batch, 0, m.nsplit,
                        m.rows_per_split, sstride, st, m.nsplit);
    if (rc) return rc;
    if (skip_dgrad) continue;   // the small-batch kernel already ran the dgrad chain
    // u is not an activation output (no mask for ly == 0)
    rc = launch_gemm<1>(l, l->int_dgrad[ly], gout, params, gin, nullptr, ly == 0 ? nullptr : hin, nullptr, batch,
                        ly == 0 ? 0 : l->act, 1, 0, 0, st);
    if (rc) return rc;
  }
  return DIB_OK;
}

int dib_integration_bwd(dib_layout* l, int batch, const float* params, float* grads, void* ws, dib_stream_t stream) {
  return integration_bwd_impl(l, batch, params, grads, ws, stream, true);
}

int dib_integration_bwd_hidden(dib_layout* l, int batch, const float* params, float* grads, void* ws, dib_stream_t stream) {
  return integration_bwd_impl(l, batch, params, grads, ws, stream, false);
}

// fused 1-unit output head of a training step: supported for out_dim == 1, linear output activation, BCE-from-logits or
// MSE, at least one integration hidden layer whose width is a multiple of 4 and <= 1024
int dib_output_head_fused_supported(const dib_layout* l, int loss_kind) {
  if (!l) return 0;
  if (!knobs().fused_head) return 0;   // dib_set_tuning("fused_head", 0): A/B switch
  if (l->out_dim != 1 || l->out_act != DIB_ACT_LINEAR || l->n_int < 1) return 0;
  if (loss_kind != DIB_LOSS_BCE_LOGITS && loss_kind != DIB_LOSS_MSE) return 0;
  const int K = l->int_width[l->n_int - 1];
  return (K % 4 == 0 && K <= 1024) ? 1 : 0;
}

int dib_output_head_fused(dib_layout* l, int loss_kind, const float* y, int64_t ldy, const int32_t* row_idx, int64_t row0,
                          int batch, float inv_global_batch, int flags, const float* params, float* grads, void* ws,
                          dib_stream_t stream) {
  const bool no_grad = (flags & DIB_HEAD_NO_GRAD) != 0;
  if (!l || !y || !params || (!grads && !no_grad) || !ws || batch <= 0) return DIB_E_ARG;
  if (!dib_output_head_fused_supported(l, loss_kind)) return DIB_E_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const auto m = l->map(batch);
  float* w = (float*)ws;
  float* gt = no_grad ? nullptr : wgrad_target(m, w, grads);
  const int ly = l->n_int, K = l->int_width[ly - 1];
  const int nblk = m.skinny_chunks, rpb = m.skinny_rows;
  const float* A = w + m.int_h[ly - 1];
  {
    ProfScope ps(kProfOther, st);
#define DIB_HEAD(NC) DIB_LAUNCH(dib_head_fused_kernel<NC>, dim3(nblk), dim3(256), 0, st, loss_kind, A, batch, K,      \
                                        params + l->int_w_off[ly], params + l->int_b_off[ly], y, (long long)ldy,                 \
                                        (const int*)row_idx, (long long)row0, inv_global_batch, l->act, rpb, w + m.pred,          \
                                        no_grad ? (float*)nullptr : w + m.g_pred, no_grad ? (float*)nullptr : w + m.g_int_h[ly - 1], \
                                        w + m.skinny_partial, w + m.loss_partial)
    if (K <= 256) DIB_HEAD(1); else if (K <= 512) DIB_HEAD(2); else DIB_HEAD(4);
#undef DIB_HEAD
    int rc = (int)hipGetLastError();
    if (rc) return rc;
    if (flags & DIB_HEAD_DEFER_SUMS) return DIB_OK;   // dib_step_tail(DIB_TAIL_HEAD_WGRAD | DIB_TAIL_LOSS_HEAD) finishes both
    if (!no_grad)
      DIB_LAUNCH(dib_skinny_wgrad_reduce_kernel, dim3(K + 1), dim3(256), 0, st, (const float*)(w + m.skinny_partial), nblk,
                         K, 1, gt + l->int_w_off[ly], gt + l->int_b_off[ly]);
    DIB_LAUNCH(dib_loss_finalize_kernel, dim3(2), dim3(256), 0, st, (const float*)(w + m.loss_partial), nblk, (float)batch,
                       w + m.step_out + l->F);
  }
  return (int)hipGetLastError();
}

static int encoder_bank_bwd_stages(dib_layout* l, int batch, const float* params, float* grads, const float* beta_dev,
                                   float inv_global_batch, int stages, void* ws, dib_stream_t stream);

// The integration network's whole share of a step with the fused 1-unit head: hidden layers forward, output Dense(1) + loss,
// and (training) the head's backward, the dgrad chain back to dL/du and the hidden layers' weight gradients.
// = dib_integration_fwd_hidden + dib_output_head_fused(flags) + dib_integration_bwd_hidden; in the row-tile regime (small_regime) the
// forward, the head and the dgrad chain are ONE launch of dib_small_integration_kernel (16-row tiles, csrc/dib_small.h).
int dib_integration_head_step(dib_layout* l, int loss_kind, const float* y, int64_t ldy, const int32_t* row_idx, int64_t row0,
                              int batch, float inv_global_batch, int flags, const float* params, float* grads, void* ws,
                              dib_stream_t stream) {
  const bool no_grad = (flags & DIB_HEAD_NO_GRAD) != 0;
  if (!l || !y || !params || (!grads && !no_grad) || !ws || batch <= 0) return DIB_E_ARG;
  if (!dib_output_head_fused_supported(l, loss_kind)) return DIB_E_UNSUPPORTED;
  if (!l->dev_groups) return DIB_E_WORKSPACE;
  int rc;
  if (use_small_int(l, batch)) {
    hipStream_t st = (hipStream_t)stream;
    const auto m = l->map(batch);
    float* w = (float*)ws;
    const int mode = DIB_SMALL_INT_FWD | DIB_SMALL_INT_HEAD |
                     (no_grad ? DIB_SMALL_INT_INFER : (DIB_SMALL_INT_HEAD_GRAD | DIB_SMALL_INT_BWD));
    rc = small_integration(l, m, w, batch, params, mode, loss_kind, y, ldy, row_idx, row0, inv_global_batch, st);
    if (rc) return rc;
    // (DIB_HEAD_DEFER_WGRAD is honoured exactly when dib_backward will run the merged weight-gradient launch: same predicate)
    if (!no_grad && !((flags & DIB_HEAD_DEFER_WGRAD) && use_merged_wgrad(l, batch))) {
      rc = integration_bwd_impl(l, batch, params, grads, ws, stream, false, /*skip_dgrad=*/true);
      if (rc) return rc;
    }
    if (flags & DIB_HEAD_DEFER_SUMS) return DIB_OK;
    const int ly = l->n_int, K = l->int_width[ly - 1];
    ProfScope ps(kProfOther, st);
    if (!no_grad) {
      float* gt = wgrad_target(m, w, grads);
      DIB_LAUNCH(dib_skinny_wgrad_reduce_kernel, dim3(K + 1), dim3(256), 0, st, (const float*)(w + m.skinny_partial),
                         m.skinny_chunks, K, 1, gt + l->int_w_off[ly], gt + l->int_b_off[ly]);
    }
    DIB_LAUNCH(dib_loss_finalize_kernel, dim3(2), dim3(256), 0, st, (const float*)(w + m.loss_partial), m.skinny_chunks,
                       (float)batch, w + m.step_out + l->F);
    return (int)hipGetLastError();
  }
  rc = integration_fwd_impl(l, batch, params, ws, stream, false);
  if (rc) return rc;
  rc = dib_output_head_fused(l, loss_kind, y, ldy, row_idx, row0, batch, inv_global_batch, flags & ~DIB_HEAD_DEFER_WGRAD, params,
                             grads, ws, stream);
  if (rc || no_grad) return rc;
  // (large batches: DIB_HEAD_DEFER_WGRAD is ignored - each layer's weight gradient runs right after its dgrad, while the
  // operands are still in the infinity cache; dib_backward(DIB_BWD_INTEGRATION_DONE) then only runs the encoder bank)
  return integration_bwd_impl(l, batch, params, grads, ws, stream, false);
}

// Everything of a step's backward pass that follows the loss, in one entry (single-GPU callers; the data-parallel bucket
// protocol keeps the separate entries): [dib_integration_bwd unless DIB_BWD_INTEGRATION_DONE] + dib_encoder_bank_bwd, with
// the weight gradients of the integration network's hidden layers that dib_integration_head_step(DIB_HEAD_DEFER_WGRAD) left
// (DIB_BWD_INTEGRATION_DONE).  In the row-tile regime ALL weight gradients of the step - encoder layers 2.., integration
// layers - are ONE grouped launch over the per-batch descriptor table dib_workspace_init wrote into the workspace.
int dib_backward(dib_layout* l, int batch, const float* params, float* grads, const float* beta_dev, float inv_global_batch,
                 int flags, void* ws, dib_stream_t stream) {
  if (!l || !params || !grads || !beta_dev || !ws || batch <= 0) return DIB_E_ARG;
  if (!l->dev_groups) return DIB_E_WORKSPACE;
  const bool int_done = (flags & DIB_BWD_INTEGRATION_DONE) != 0;
  const bool merged = use_merged_wgrad(l, batch) && (int_done || l->out_dim % 16 == 0);
  int rc;
  if (!merged) {
    // (int_done here means dib_integration_head_step already ran the integration network's weight gradients: it defers them
    // only under the predicate that makes `merged` true)
    rc = int_done ? DIB_OK : integration_bwd_impl(l, batch, params, grads, ws, stream, true);
    if (rc) return rc;
    return encoder_bank_bwd_stages(l, batch, params, grads, beta_dev, inv_global_batch, 3, ws, stream);
  }
  hipStream_t st = (hipStream_t)stream;
  const auto m = l->map(batch);
  float* w = (float*)ws;
  if (!int_done) {   // dgrad chain from ws[G_PRED] (general output layer) down to ws[G_U]
    rc = small_integration(l, m, w, batch, params, DIB_SMALL_INT_LOAD_H | DIB_SMALL_INT_BWD_OUT | DIB_SMALL_INT_BWD, 0, nullptr, 0,
                           nullptr, 0, 0.f, st);
    if (rc) return rc;
  }
  rc = small_encoder_bwd(l, m, w, batch, params, beta_dev, inv_global_batch, st);
  if (rc) return rc;
  // encoder layers 1 .. n_enc (layer 0 comes out of the backward kernel as partials), integration hidden layers, and the
  // general output layer when this call ran its backward (the fused head reduces its own partials in the tail)
  const int count = l->n_enc * l->F + l->n_int + (int_done ? 0 : 1);
  return merged_wgrad(l, m, w, batch, wgrad_target(m, w, grads), 0, count, st);
}


// stages: bit 0 = the gradient chain (reparam/KL backward + dgrads) and every weight gradient except the last encoder
// layer's; bit 1 = the last layer's weight gradient (independent of the others: it reads dout and the last hidden layer).
// 3 = both, in the single-GPU order (last layer first).  The data-parallel caller runs stage 1, finalizes + all-reduces
// part 2 (the front layers), then runs stage 2 under that all-reduce (dib_encoder_bank_bwd_stage).
static int encoder_bank_bwd_stages(dib_layout* l, int batch, const float* params, float* grads, const float* beta_dev,
                                   float inv_global_batch, int stages, void* ws, dib_stream_t stream) {
  if (!l || !params || !grads || !beta_dev || !ws || batch <= 0 || stages < 1 || stages > 3) return DIB_E_ARG;
  if (!l->dev_groups) return DIB_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const auto m = l->map(batch);
  float* w = (float*)ws;
  float* gt = wgrad_target(m, w, grads);
  const long long sstride = align_up(l->n_params, 4);
  int rc = DIB_OK;
  const bool small = use_small_enc(l, batch);
  const bool fused = small || fused_bwd_ok(l);   // d(W1|b1) comes as per-workgroup partials, dgrads in one launch
  const int LE = l->n_enc + 1;
  if (stages & 1) {
    if (small) {  // 16-row tiles (dib_small.h)
      rc = small_encoder_bwd(l, m, w, batch, params, beta_dev, inv_global_batch, st);
    } else if (fused) {  // reparam/KL backward + both dgrads in one launch (dib_fused.h); wgrads below read its outputs
      rc = fused_encoder_bwd(l, m, w, batch, params, beta_dev, inv_global_batch, st);
    } else {
      { ProfScope ps(kProfOther, (hipStream_t)stream);
      DIB_LAUNCH(dib_reparam_kl_bwd_kernel, dim3(m.kl_blocks, l->F), dim3(256), 0, st, w + m.enc_out, w + m.g_u,
                         w + m.U, w + m.dout, beta_dev, inv_global_batch, batch, l->F, l->E); }
      rc = (int)hipGetLastError();
    }
    if (rc) return rc;
  }
  for (int ly = LE - 1; ly >= 0; --ly) {
    const bool last = ly == LE - 1;
    const float* gout = last ? w + m.dout : w + m.g_enc_h[ly];
    const float* hin = ly == 0 ? w + m.P : w + m.enc_h[ly - 1];
    const bool wgrad_here = (last ? (stages & 2) : (stages & 1)) && !(fused && ly == 0);  // fused: d(W1|b1) comes out of the
                                                                                         // fused kernel, reduced at finalize
    if (wgrad_here) {
      // narrow outputs (the 2E-wide last layer) run 128x64 tiles at 4 workgroups/CU: half as many, twice as long batch
      // splits fill the chip in one wave (measured 0.88 -> 0.71 ms); the unused slabs of these blocks stay zero.
      // (only from 32 splits = 16384 rows up: at B = 8192 the 16 -> 8 split halving measured 117 vs 103 us)
      // (the per-launch split rule, pick_wgrad_splits, starts from this choice and leaves it unless it predicts > 5 % better)
      const bool halve = l->enc_wgrad[ly].max_n <= 64 && m.nsplit >= 32 && (m.nsplit % 2) == 0;
      rc = launch_gemm<2>(l, l->enc_wgrad[ly], hin, gout, gt, nullptr, nullptr, gt, batch, 0,
                          halve ? m.nsplit / 2 : m.nsplit, halve ? 2 * m.rows_per_split : m.rows_per_split, sstride, st,
                          m.nsplit);
      if (rc) return rc;
    }
    if ((stages & 1) && ly >= 1 && !fused) {
      rc = launch_gemm<1>(l, l->enc_dgrad[ly], gout, params, w + m.g_enc_h[ly - 1], nullptr, hin, nullptr, batch,
                          l->act, 1, 0, 0, st);
      if (rc) return rc;
    }
  }
  return DIB_OK;
}

int dib_encoder_bank_bwd(dib_layout* l, int batch, const float* params, float* grads, const float* beta_dev,
                         float inv_global_batch, void* ws, dib_stream_t stream) {
  return encoder_bank_bwd_stages(l, batch, params, grads, beta_dev, inv_global_batch, 3, ws, stream);
}

int dib_encoder_bank_bwd_stage(dib_layout* l, int batch, const float* params, float* grads, const float* beta_dev,
                               float inv_global_batch, int stage, void* ws, dib_stream_t stream) {
  if (stage != 1 && stage != 2) return DIB_E_ARG;
  return encoder_bank_bwd_stages(l, batch, params, grads, beta_dev, inv_global_batch, stage, ws, stream);
}

// [beg, end) of a gradient bucket in the flat buffers.  The layout is layer-major (all features' kernels of encoder layer 0,
// their biases, layer 1, ..., then the integration network), every block boundary a multiple of 4 floats:
//   0 = encoder bank, 1 = integration network, 2 = encoder layers before the last ("front"), 3 = last encoder layer ("tail"),
//   -1 = everything
static void part_bounds(const dib_layout* l, int part, long long* beg, long long* end) {
  const long long split = l->int_w_off[0], tail = l->enc_w_off[l->n_enc][0], all = l->n_params;
  switch (part) {
    case 0: *beg = 0; *end = split; break;
    case 1: *beg = split; *end = all; break;
    case 2: *beg = 0; *end = tail; break;
    case 3: *beg = tail; *end = split; break;
    default: *beg = 0; *end = all; break;
  }
}

int dib_layout_part_range(const dib_layout* l, int part, int64_t* offset, int64_t* count) {
  if (!l || !offset || !count || part < 0 || part > 3) return DIB_E_ARG;
  long long beg, end;
  part_bounds(l, part, &beg, &end);
  *offset = beg;
  *count = end - beg;
  return DIB_OK;
}

// part: see part_bounds
int dib_grads_finalize_part(dib_layout* l, int batch, int part, float* grads, void* ws, dib_stream_t stream) {
  if (!l || !grads || !ws || batch <= 0 || part < -1 || part > 3) return DIB_E_ARG;
  const auto m = l->map(batch);
  hipStream_t st = (hipStream_t)stream;
  float* w = (float*)ws;
  if (m.nsplit > 1) {
    // partial slabs are spaced align_up(n_params,4) apart
    const long long stride = align_up(l->n_params, 4);
    long long beg, end;
    part_bounds(l, part, &beg, &end);
    if (part == -1 || part == 1) end = stride;   // the last bucket carries the alignment tail of the buffer
    { ProfScope ps(kProfOther, (hipStream_t)stream);
    DIB_LAUNCH(dib_reduce_splits_kernel, dim3(grid_for((end - beg) / 4)), dim3(256), 0, st,
                       (const float*)(w + m.wgrad_partial + beg), end - beg, m.nsplit, stride, grads + beg); }
    int rc = (int)hipGetLastError();
    if (rc) return rc;
  }
  if (part != 1 && part != 3 && enc_dw1_parts(l, batch) > 0) {  // layer-1 weight/bias gradients: fixed-order sum of the backward's partials
    ProfScope ps(kProfOther, (hipStream_t)stream);
    DIB_LAUNCH(dib_dw1_reduce_kernel, dim3(l->F, 16), dim3(256), 0, st, (const float*)(w + m.dw1_partial),
                       enc_dw1_parts(l, batch), l->F, l->enc_units[0], l->dev_fused_offs, l->dev_fused_offs + 3 * l->F,
                       l->dev_featmap, grads);
  }
  return (int)hipGetLastError();
}

int dib_grads_finalize(dib_layout* l, int batch, float* grads, void* ws, dib_stream_t stream) {
  return dib_grads_finalize_part(l, batch, -1, grads, ws, stream);
}

int dib_metrics_accumulate(dib_layout* l, int batch, const float* beta_dev, float inv_global_batch,
                           float* metrics_acc, void* ws, dib_stream_t stream) {
  if (!l || !beta_dev || !metrics_acc || !ws || batch <= 0) return DIB_E_ARG;
  const auto m = l->map(batch);
  float* w = (float*)ws;
  { ProfScope ps(kProfOther, (hipStream_t)stream);
  DIB_LAUNCH(dib_metrics_accumulate_kernel, dim3(cdiv(l->F + 3, 64)), dim3(64), 0, (hipStream_t)stream,
                     w + m.step_out, l->F, beta_dev, inv_global_batch, metrics_acc); }
  return (int)hipGetLastError();
}

// ---- the end of a step in one launch (csrc/dib_tail.h) ------------------------------------------------------------
int dib_step_tail(dib_layout* l, int batch, int part, int flags, float* params, float* grads, float* adam_m, float* adam_v,
                  const float* lr_dev, int64_t* t_dev, float beta1, float beta2, float eps, float grad_scale,
                  const float* beta_dev, float inv_global_batch, float* metrics_acc, void* ws, dib_stream_t stream) {
  if (!l || !ws || batch <= 0 || part < -1 || part > 3 || flags <= 0) return DIB_E_ARG;
  const bool adam = (flags & DIB_TAIL_ADAM) != 0, sgd = (flags & DIB_TAIL_SGD) != 0, finalize = (flags & DIB_TAIL_FINALIZE) != 0;
  if (adam && sgd) return DIB_E_ARG;
  if ((finalize || adam || sgd || (flags & DIB_TAIL_HEAD_WGRAD)) && !grads) return DIB_E_ARG;
  if ((adam || sgd) && (!params || !lr_dev)) return DIB_E_ARG;
  if (adam && (!adam_m || !adam_v || !t_dev)) return DIB_E_ARG;
  if ((flags & DIB_TAIL_BUMP) && !t_dev) return DIB_E_ARG;
  if ((flags & DIB_TAIL_METRICS) && (!beta_dev || !metrics_acc)) return DIB_E_ARG;
  if ((flags & DIB_TAIL_LOSS) && (flags & DIB_TAIL_LOSS_HEAD)) return DIB_E_ARG;
  const auto m = l->map(batch);
  float* w = (float*)ws;
  const long long stride = align_up(l->n_params, 4);
  DibTailArgs a;
  std::memset(&a, 0, sizeof(a));
  a.params = params; a.grads = grads; a.m = adam_m; a.v = adam_v; a.lr_dev = lr_dev; a.t_dev = (long long*)t_dev;
  a.b1 = beta1; a.b2 = beta2; a.eps = eps; a.gscale = grad_scale; a.flags = flags; a.F = l->F;
  const bool touches = finalize || adam || sgd;   // this launch walks the part's gradient range
  long long beg = 0, end = 0;
  part_bounds(l, part, &beg, &end);
  if (part == -1 || part == 1) end = stride;      // the last bucket carries the alignment tail of the buffer
  const bool dw1_seg = finalize && enc_dw1_parts(l, batch) > 0 && part != 1 && part != 3;
  const bool head_seg = (flags & DIB_TAIL_HEAD_WGRAD) && (part == -1 || part == 1);
  if (head_seg && !dib_output_head_fused_supported(l, DIB_LOSS_BCE_LOGITS)) return DIB_E_UNSUPPORTED;
  if (touches) {
    a.gbeg = dw1_seg ? l->enc_w_off[1][0] : beg;
    a.gend = head_seg ? l->int_w_off[l->n_int] : end;
    if (finalize && m.nsplit > 1) { a.slabs = w + m.wgrad_partial; a.nsplit = m.nsplit; a.slab_stride = stride; }
    const long long n4 = (a.gend - a.gbeg) >> 2;
    if (n4 > 0 && (a.nsplit > 0 || adam || sgd)) a.nb_generic = (int)std::min<long long>(2048, (n4 + 255) / 256);
  }
  if (dw1_seg) {
    a.dw1_partial = w + m.dw1_partial; a.dw1_parts = enc_dw1_parts(l, batch); a.H1 = l->enc_units[0];
    a.w_off = l->dev_fused_offs; a.b_off = l->dev_fused_offs + 3 * l->F; a.featmap = l->dev_featmap;
    a.nb_dw1 = l->F * 16;
  }
  if (head_seg) {
    a.head_partial = w + m.skinny_partial; a.head_chunks = m.skinny_chunks; a.head_K = l->int_width[l->n_int - 1];
    a.head_w_off = l->int_w_off[l->n_int]; a.head_b_off = l->int_b_off[l->n_int];
    a.nb_head = a.head_K + 1;
  }
  a.step_out = w + m.step_out;
  if (flags & DIB_TAIL_KL) {
    a.kl_partial = w + m.kl_partial; a.kl_stride = l->F; a.nb_kl = l->F;
    a.kl_rows = enc_kl_rows(l, m, batch);
  }
  if (flags & (DIB_TAIL_LOSS | DIB_TAIL_LOSS_HEAD)) {
    a.loss_partial = w + m.loss_partial; a.nb_loss = 2; a.rows = (float)batch;
    a.loss_blocks = (flags & DIB_TAIL_LOSS_HEAD) ? m.skinny_chunks : m.loss_blocks;
  }
  a.beta_dev = beta_dev; a.inv_bg = inv_global_batch; a.metrics_acc = metrics_acc;
  a.sync = (unsigned*)(w + m.sync);
  int grid = a.nb_generic + a.nb_dw1 + a.nb_head + a.nb_kl + a.nb_loss;
  if (grid == 0 && !(flags & (DIB_TAIL_BUMP | DIB_TAIL_METRICS))) return DIB_OK;   // nothing to reduce, nothing to step
  grid = std::max(1, grid);
  ProfScope ps(kProfOther, (hipStream_t)stream);
  DIB_LAUNCH(dib_step_tail_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

// ---- tuning: the one documented switchboard (no environment variables) -------------------------------------------
static int* tuning_slot(const char* key) {
  Tuning& t = tuning();
  if (!key) return nullptr;
  if (!std::strcmp(key, "fwd_small_wgs")) return &t.fwd_small_wgs;
  if (!std::strcmp(key, "fwd_narrow_wgs")) return &t.fwd_narrow_wgs;
  if (!std::strcmp(key, "stream_rows")) return &t.stream_rows;
  if (!std::strcmp(key, "split_policy")) return &t.split_policy;
  if (!std::strcmp(key, "split_overhead")) return &t.split_overhead;
  if (!std::strcmp(key, "fused_encoder")) return &t.fused_encoder;
  if (!std::strcmp(key, "fused_head")) return &t.fused_head;
  if (!std::strcmp(key, "small_batch")) return &t.small_batch;
  if (!std::strcmp(key, "small_wgs")) return &t.small_wgs;
  if (!std::strcmp(key, "mlp_row_tiles")) return &t.mlp_row_tiles;
  if (!std::strcmp(key, "infonce_one_launch")) return &t.infonce_one_launch;
  if (!std::strcmp(key, "attn_small_bwd_waves")) return &t.attn_small_bwd_waves;
  if (!std::strcmp(key, "wgrad_flat_tile")) return &t.wgrad_flat_tile;
  if (!std::strcmp(key, "wgrad_max_splits")) return &t.wgrad_max_splits;
  if (!std::strcmp(key, "num_cus")) return &t.num_cus;
  if (!std::strcmp(key, "attn_fwd_waves")) return &t.attn_fwd_waves;
  if (!std::strcmp(key, "int_cluster_short_exchange")) return &t.int_cluster_short_exchange;
  if (!std::strcmp(key, "int_cluster")) return &t.int_cluster;
  if (!std::strcmp(key, "int_cluster_wgs")) return &t.int_cluster_wgs;
  if (!std::strcmp(key, "int_cluster_min_weights")) return &t.int_cluster_min_weights;
  return nullptr;
}

int dib_set_tuning(const char* key, int value) {
  int* p = tuning_slot(key);
  if (!p || value < 0) return DIB_E_ARG;
  *p = value;
  return DIB_OK;
}

int dib_get_tuning(const char* key, int* value) {
  const int* p = tuning_slot(key);
  if (!p || !value) return DIB_E_ARG;
  *value = *p;
  return DIB_OK;
}

// ---- optimizers ------------------------------------------------------------------------------------
int dib_adam_step(float* params, const float* grads, float* mm, float* vv, int64_t n, const float* lr_dev,
                  int64_t* t_dev, float beta1, float beta2, float eps, float grad_scale, dib_stream_t stream) {
  if (!params || !grads || !mm || !vv || !lr_dev || !t_dev || n <= 0) return DIB_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  { ProfScope ps(kProfOther, (hipStream_t)stream);
  DIB_LAUNCH(dib_adam_kernel, dim3(grid_for(n / 4 + 1)), dim3(256), 0, st, params, grads, mm, vv, (long long)n,
                     lr_dev, (const long long*)t_dev, beta1, beta2, eps, grad_scale); }
  int rc = (int)hipGetLastError();
  if (rc) return rc;
  { ProfScope ps(kProfOther, (hipStream_t)stream);
  DIB_LAUNCH(dib_bump_counter_kernel, dim3(1), dim3(1), 0, st, (long long*)t_dev); }
  return (int)hipGetLastError();
}

int dib_sgd_step(float* params, const float* grads, int64_t n, const float* lr_dev, float grad_scale,
                 dib_stream_t stream) {
  if (!params || !grads || !lr_dev || n <= 0) return DIB_E_ARG;
  { ProfScope ps(kProfOther, (hipStream_t)stream);
  DIB_LAUNCH(dib_sgd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, params, grads, (long long)n,
                     lr_dev, grad_scale); }
  return (int)hipGetLastError();
}

// ---- evaluation helpers ------------------------------------------------------------------------------
int dib_encode_deterministic(dib_layout* l, int feature, const float* x_f, int n, const float* params, float* out,
                             void* ws, dib_stream_t stream) {
  if (!l || !x_f || !params || !out || !ws || n <= 0) return DIB_E_ARG;
  if (feature < 0 || feature >= l->F) return DIB_E_ARG;
  if (!l->dev_groups) return DIB_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const auto m = l->map(n);
  float* w = (float*)ws;
  const int d = l->dims[feature];
  { ProfScope ps(kProfOther, (hipStream_t)stream);
  DIB_LAUNCH(dib_posenc_kernel<64>, dim3(cdiv(d, 64), cdiv(n, 64)), dim3(256), 0, st, x_f, (long long)d,
                     (const int*)nullptr, 0ll, n, l->dev_colmap + l->x_off[feature], d, l->n_blocks, w + m.P); }
  int rc = (int)hipGetLastError();
  if (rc) return rc;
  rc = encoder_chain_fwd(l, m, w, n, params, feature, 1, st);
  if (rc) return rc;
  const int w2 = 2 * l->E;
  return (int)hipMemcpyAsync(out, w + m.enc_out + (int64_t)feature * n * w2, (size_t)n * w2 * sizeof(float),
                             hipMemcpyDeviceToDevice, st);
}

int dib_bhattacharyya(const float* mu1, const float* lv1, int n, const float* mu2, const float* lv2, int m, int dim,
                      float* out, dib_stream_t stream) {
  if (!mu1 || !lv1 || !mu2 || !lv2 || !out || n <= 0 || m <= 0 || dim <= 0) return DIB_E_ARG;
  DIB_LAUNCH(dib_bhattacharyya_kernel, dim3(grid_for((int64_t)n * m)), dim3(256), 0, (hipStream_t)stream, mu1,
                     lv1, n, mu2, lv2, m, dim, out);
  return (int)hipGetLastError();
}

int64_t dib_infonce_workspace_bytes(int batch) {
  if (batch <= 0) return DIB_E_ARG;
  // VALU path (l1, linf): S, ST, C, CT, C2, C2T [B^2 floats each] | arg-max, its transpose [B^2 int32 each] | lse [2B] | norms [2B]
  // MFMA path (l2sq, l2, cosine; csrc/dib_infonce_mfma.h): S [B^2] | 32-wide block partials of the row / column log-sum-exp
  // [4 ceil(B/32) B] inside the same 8 B^2 | lse | norms | slice partials of C . Other [2 x 8 x B x 256 at most] and of the row
  // sums [2 x 8 x B]
  return (int64_t)sizeof(float) * (8ll * batch * batch + 4ll * batch + 64 + (4096ll + 16ll) * batch + 64);
}

int dib_infonce_fwd_bwd(const float* emb_x, const float* emb_y, int batch, int dim, int similarity, float temperature,
                        float* g_x, float* g_y, float* loss_out, void* ws, dib_stream_t stream) {
  if (!emb_x || !emb_y || !loss_out || !ws || batch <= 0 || dim <= 0 || temperature <= 0.f) return DIB_E_ARG;
  if (similarity < 0 || similarity > 4 || dim > 256) return DIB_E_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const int64_t bb = (int64_t)batch * batch;
  float* S = (float*)ws;
  float* ST = S + bb;
  float* C = ST + bb;
  float* CT = C + bb;
  float* C2 = CT + bb;
  float* C2T = C2 + bb;
  int* amax = (int*)(C2T + bb);
  int* amaxT = amax + bb;
  float* lse = (float*)(amaxT + bb);
  float* norms = lse + 2ll * batch;
  const float inv_t = 1.0f / temperature;
  const int tiles = cdiv(batch, 32);
  static std::atomic<bool> attr_set[64];
  if (AttrOnce once(attr_set); once) {   // two 32-row tiles of up to 256 (+1) floats: 65 792 bytes at the widest
    hipError_t e = hipFuncSetAttribute((const void*)dib_infonce_sim_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       2 * 32 * 257 * (int)sizeof(float));
    if (e != hipSuccess) return (int)e;
  }
  ProfScope ps(kProfOther, st);
  if ((similarity == 0 || similarity == 1 || similarity == 4) && batch <= DIB_INCE1_MAXB && dim <= 64 && knobs().infonce_one_launch) {
    // the reference's default batch: similarity, log-sum-exps, loss and both gradients in ONE launch (dib_infonce_small_kernel)
    const bool grads = g_x && g_y;
    const dim3 grid(grads ? cdiv(batch, 64) : 1, grads ? 2 : 1);
    const size_t lds = (size_t)DIB_INCE1_LDS_FLOATS * sizeof(float);
    static int lds_have[3][64] = {};
#define DIB_INCE_ONE(KD, SLOT)                                                                                                  \
    do {                                                                                                                        \
      if (int rc = ensure_dynamic_lds((const void*)dib_infonce_small_kernel<KD>, lds, lds_have[SLOT])) return rc;               \
      DIB_LAUNCH(dib_infonce_small_kernel<KD>, grid, dim3(DIB_INCE1_THREADS), lds, st, emb_x, emb_y, batch, dim, inv_t, temperature,          \
                 grads ? g_x : (float*)nullptr, grads ? g_y : (float*)nullptr, loss_out);                                       \
    } while (0)
    if (similarity == 0) DIB_INCE_ONE(0, 0); else if (similarity == 1) DIB_INCE_ONE(1, 1); else DIB_INCE_ONE(4, 2);
#undef DIB_INCE_ONE
    return (int)hipGetLastError();
  }
  if (similarity == 0 || similarity == 1 || similarity == 4) {
    // dot-product similarities: S, g_x = C Y, g_y = C^T X as three MFMA products (csrc/dib_infonce_mfma.h)
    const int nb32 = cdiv(batch, 32), t64 = cdiv(batch, 64);
    float* prow = S + bb;
    float* pcol = prow + 2ll * nb32 * batch;
    unsigned* arrive = (unsigned*)(norms + 2ll * batch + 16);
    float* Gp = norms + 2ll * batch + 64;
    // partner slices per (self block, side): one up to B = 256 (the gradient kernel then writes g itself: 3 launches in all);
    // above, enough workgroups to fill 256 CUs twice, at most 8 (the partial buffers' size)
    const int nsplit = t64 <= 4 ? 1 : std::max(1, std::min(std::min(t64, 8), cdiv(512, 2 * t64)));
    float* Rp = Gp + 2ll * nsplit * batch * dim;
    const int nacc = cdiv(dim, 64);
    const size_t os_bytes = (size_t)64 * (64 * nacc + 4) * sizeof(float);
    static std::atomic<bool> attr_mfma[64];
    if (AttrOnce once(attr_mfma); once) {
#define DIB_INCE_ATTR(NA, KD)                                                                                          \
      if (hipFuncSetAttribute((const void*)dib_infonce_grad_mfma_kernel<NA, KD>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                              64 * (64 * NA + 4) * (int)sizeof(float)) != hipSuccess) return DIB_E_ARG;
      DIB_INCE_ATTR(3, 0) DIB_INCE_ATTR(3, 1) DIB_INCE_ATTR(3, 4) DIB_INCE_ATTR(4, 0) DIB_INCE_ATTR(4, 1) DIB_INCE_ATTR(4, 4)
#undef DIB_INCE_ATTR
    }
#define DIB_INCE_SIM(KD) DIB_LAUNCH(dib_infonce_sim_mfma_kernel<KD>, dim3(t64, t64), dim3(256), 0, st, emb_x, emb_y, batch, \
                                            dim, inv_t, norms, S, prow, pcol, nb32, arrive)
    if (similarity == 0) DIB_INCE_SIM(0); else if (similarity == 1) DIB_INCE_SIM(1); else DIB_INCE_SIM(4);
#undef DIB_INCE_SIM
    float* lpart = pcol + 2ll * nb32 * batch;   // one loss partial per lse workgroup (still inside the 8 B^2 region)
    DIB_LAUNCH(dib_infonce_lse_loss_kernel, dim3(cdiv(2 * batch, 32)), dim3(256), 0, st, (const float*)prow,
                       (const float*)pcol, (const float*)S, batch, nb32, lse, arrive, lpart, loss_out);
    if (g_x && g_y) {
      const dim3 grid(t64, nsplit, 2);
#define DIB_INCE_GRAD(NA, KD) DIB_LAUNCH((dib_infonce_grad_mfma_kernel<NA, KD>), grid, dim3(256), os_bytes, st, emb_x, emb_y, \
                                                 (const float*)S, (const float*)lse, (const float*)norms, batch, dim, inv_t,            \
                                                 temperature, nsplit, Gp, Rp, g_x, g_y)
#define DIB_INCE_GRAD_K(KD) do { if (nacc == 1) DIB_INCE_GRAD(1, KD); else if (nacc == 2) DIB_INCE_GRAD(2, KD);              \
                                 else if (nacc == 3) DIB_INCE_GRAD(3, KD); else DIB_INCE_GRAD(4, KD); } while (0)
      if (similarity == 0) DIB_INCE_GRAD_K(0); else if (similarity == 1) DIB_INCE_GRAD_K(1); else DIB_INCE_GRAD_K(4);
#undef DIB_INCE_GRAD_K
#undef DIB_INCE_GRAD
      if (nsplit > 1)
        DIB_LAUNCH(dib_infonce_grad_final_kernel, dim3(cdiv(2ll * batch * dim, 256)), dim3(256), 0, st, emb_x, emb_y,
                           (const float*)Gp, (const float*)Rp, batch, dim, similarity, nsplit, g_x, g_y);
    }
    return (int)hipGetLastError();
  }
  DIB_LAUNCH(dib_infonce_sim_kernel, dim3(tiles, tiles), dim3(256), (size_t)2 * 32 * (dim + 1) * sizeof(float), st, emb_x,
                     emb_y, batch, dim, similarity, inv_t, (const float*)norms, S, ST, amax, amaxT);
  DIB_LAUNCH(dib_infonce_lse_kernel, dim3(batch, 2), dim3(256), 0, st, (const float*)S, (const float*)ST, batch, lse);
  DIB_LAUNCH(dib_infonce_loss_kernel, dim3(1), dim3(256), 0, st, (const float*)S, (const float*)lse, batch,
                     loss_out);
  if (g_x && g_y) {
    DIB_LAUNCH(dib_infonce_coef_kernel, dim3(grid_for(bb, 256, 2048), 2), dim3(256), 0, st, (const float*)S,
                       (const float*)ST, (const float*)lse, (const float*)norms, batch, similarity, inv_t, temperature, C, CT, C2,
                       C2T);
    DIB_LAUNCH(dib_infonce_grad_kernel, dim3(batch, 2), dim3(256), 256 * sizeof(float), st, emb_x, emb_y,
                       (const float*)C, (const float*)CT, (const float*)C2, (const float*)C2T, (const int*)amax,
                       (const int*)amaxT, batch, dim, similarity, g_x, g_y);
  }
  return (int)hipGetLastError();
}

int dib_positional_encoding(const float* x, int64_t ldx, int n, int d, int n_freq, float* out, dib_stream_t stream) {
  if (!x || !out || n <= 0 || d <= 0) return DIB_E_ARG;
  const int n_blocks = n_freq > 1 ? n_freq : 1;
  DIB_LAUNCH(dib_posenc_dense_kernel, dim3(grid_for((int64_t)n * d)), dim3(256), 0, (hipStream_t)stream, x,
                     (long long)ldx, n, d, n_blocks, out);
  return (int)hipGetLastError();
}

int dib_positional_encoding_rows(const float* x, int64_t ldx, const int32_t* row_idx, int n, int d, int n_freq, float* out,
                                 dib_stream_t stream) {
  if (!x || !row_idx || !out || n <= 0 || d <= 0) return DIB_E_ARG;
  const int n_blocks = n_freq > 1 ? n_freq : 1;
  DIB_LAUNCH(dib_posenc_rows_kernel, dim3(grid_for((int64_t)n * d)), dim3(256), 0, (hipStream_t)stream, x, (long long)ldx,
                     (const int*)row_idx, n, d, n_blocks, out);
  return (int)hipGetLastError();
}

// ---- plain MLP on the row-tile kernels (include/dib_hip.h dib_mlp_small_*): dib_small_integration_kernel with its input
// tile built from the batch's rows of X (DIB_SMALL_INT_POSENC_IN) and the dgrad chain stopped at the first layer ----
static int64_t mlp_small_lds_floats(const dib_mlp_desc* d) {
  const int nf = d->n_freq > 1 ? d->n_freq : 1;
  int64_t fl = (int64_t)DIB_SMALL_ROWS * dib_small_pitch(d->in_dim * nf);
  for (int i = 0; i < d->n_hidden; ++i) fl += 2ll * DIB_SMALL_ROWS * dib_small_pitch(d->width[i]);
  return fl + (int64_t)DIB_SMALL_ROWS * dib_small_pitch(d->width[d->n_hidden]) + DIB_SMALL_XCH_FLOATS;
}
int dib_mlp_small_supported(const dib_mlp_desc* d, int batch) {
  if (!d || !knobs().small_batch || !knobs().mlp_row_tiles || batch < 1 || batch > kSmallMaxBatch) return 0;
  if (d->n_hidden < 1 || d->n_hidden > 3 || d->in_dim < 1) return 0;
  if (!(d->act >= 0 && d->act <= 2) && d->act != DIB_ACT_LEAKY_RELU_01) return 0;   // piecewise-linear activations only
  const int nf = d->n_freq > 1 ? d->n_freq : 1;
  if ((int64_t)d->in_dim * nf > 1024) return 0;
  for (int i = 0; i <= d->n_hidden; ++i)
    if (d->width[i] < 16 || d->width[i] % 16 != 0 || d->width[i] > 1024) return 0;
  return mlp_small_lds_floats(d) * 4 <= 150 * 1024 ? 1 : 0;
}
static void mlp_small_fill(const dib_mlp_desc* d, DibSmallIntArgs& a, const float* params, int n) {
  const int nf = d->n_freq > 1 ? d->n_freq : 1;
  a.batch = n; a.K0 = d->in_dim * nf; a.params = params; a.n_hidden = d->n_hidden;
  for (int i = 0; i <= d->n_hidden; ++i) { a.width[i] = d->width[i]; a.w_off[i] = d->w_off[i]; a.b_off[i] = d->b_off[i]; }
  a.act = d->act; a.out_act = 0; a.out_dim = d->width[d->n_hidden];
  a.in_dim = d->in_dim; a.n_freq = nf;
}
static int mlp_small_launch(const dib_mlp_desc* d, const DibSmallIntArgs& a, hipStream_t st) {
  const size_t lds = (size_t)mlp_small_lds_floats(d) * 4;
  static int lds_have[64] = {};
  if (int rc = ensure_dynamic_lds((const void*)dib_small_integration_kernel, lds, lds_have)) return rc;
  ProfScope ps(kProfOther, st);
  DIB_LAUNCH(dib_small_integration_kernel, dim3(small_tiles(a.batch)), dim3(DIB_SMALL_THREADS), lds, st, a);
  return (int)hipGetLastError();
}
// argument sets of the two passes (validated); DIB_OK, or the error the stand-alone entry point reports
static int mlp_small_fwd_args(const dib_mlp_desc* d, const float* params, const float* x, int64_t ldx, const int32_t* row_idx, int n,
                              float* a0, float* const* h, float* out, DibSmallIntArgs& a) {
  if (!d || !params || !x || !out || n <= 0) return DIB_E_ARG;
  if (!dib_mlp_small_supported(d, n)) return DIB_E_UNSUPPORTED;
  bool stash = a0 != nullptr;
  for (int i = 0; i < d->n_hidden; ++i) stash = stash && h != nullptr && h[i] != nullptr;
  if (a0 != nullptr && !stash) return DIB_E_ARG;
  std::memset(&a, 0, sizeof(a));
  a.mode = DIB_SMALL_INT_FWD | DIB_SMALL_INT_OUT | DIB_SMALL_INT_POSENC_IN | (stash ? 0 : DIB_SMALL_INT_INFER);
  a.X = x; a.ldx = ldx; a.row_idx = (const int*)row_idx; a.row0 = 0; a.a0 = a0; a.pred = out;
  if (stash) for (int i = 0; i < d->n_hidden; ++i) a.h[i] = h[i];
  mlp_small_fill(d, a, params, n);
  return DIB_OK;
}
static int mlp_small_bwd_args(const dib_mlp_desc* d, const float* params, const float* g_out, float* const* h, float* const* g, int n,
                              DibSmallIntArgs& a) {
  if (!d || !params || !g_out || !h || !g || n <= 0) return DIB_E_ARG;
  if (!dib_mlp_small_supported(d, n)) return DIB_E_UNSUPPORTED;
  std::memset(&a, 0, sizeof(a));
  a.mode = DIB_SMALL_INT_LOAD_H | DIB_SMALL_INT_BWD_OUT | DIB_SMALL_INT_BWD | DIB_SMALL_INT_NO_GU;
  a.g_pred = const_cast<float*>(g_out);
  for (int i = 0; i < d->n_hidden; ++i) {
    if (!h[i] || !g[i]) return DIB_E_ARG;
    a.h[i] = h[i]; a.g[i] = g[i];
  }
  mlp_small_fill(d, a, params, n);
  return DIB_OK;
}
int dib_mlp_small_fwd(const dib_mlp_desc* d, const float* params, const float* x, int64_t ldx, const int32_t* row_idx, int n,
                      float* a0, float* const* h, float* out, dib_stream_t stream) {
  if (n == 0) return DIB_OK;
  DibSmallIntArgs a;
  if (int rc = mlp_small_fwd_args(d, params, x, ldx, row_idx, n, a0, h, out, a)) return rc;
  return mlp_small_launch(d, a, (hipStream_t)stream);
}
int dib_mlp_small_bwd(const dib_mlp_desc* d, const float* params, const float* g_out, float* const* h, float* const* g, int n,
                      dib_stream_t stream) {
  if (n == 0) return DIB_OK;
  DibSmallIntArgs a;
  if (int rc = mlp_small_bwd_args(d, params, g_out, h, g, n, a)) return rc;
  return mlp_small_launch(d, a, (hipStream_t)stream);
}
// ---- plain MLP with a 1-unit head: the whole training step of the head network in ONE launch (include/dib_hip.h) ----
static int64_t mlp_head_lds_floats(const dib_mlp_desc* d) {
  int64_t fl = (int64_t)DIB_SMALL_ROWS * dib_small_pitch(d->in_dim);
  for (int i = 0; i < d->n_hidden; ++i) fl += 2ll * DIB_SMALL_ROWS * dib_small_pitch(d->width[i]);
  return fl + (int64_t)DIB_SMALL_ROWS * dib_small_pitch(1) + DIB_SMALL_XCH_FLOATS + 9 * (d->width[d->n_hidden - 1] + 1) + 32;
}
int dib_mlp_small_head_supported(const dib_mlp_desc* d, int n) {
  if (!d || !knobs().small_batch || !knobs().mlp_row_tiles || n < 1 || n > kSmallMaxBatch) return 0;
  if (d->n_hidden < 1 || d->n_hidden > 3 || d->in_dim < 16 || d->in_dim % 16 || d->in_dim > 1024 || d->n_freq > 1) return 0;
  if (!(d->act >= 0 && d->act <= 2) && d->act != DIB_ACT_LEAKY_RELU_01) return 0;
  for (int i = 0; i < d->n_hidden; ++i)
    if (d->width[i] < 16 || d->width[i] % 16 != 0 || d->width[i] > 1024) return 0;
  if (d->width[d->n_hidden] != 1) return 0;
  return mlp_head_lds_floats(d) * 4 <= 150 * 1024 ? 1 : 0;
}
int64_t dib_mlp_small_head_workspace_bytes(const dib_mlp_desc* d, int n) {
  if (!d || n < 1 || d->n_hidden < 1 || d->n_hidden > 3) return DIB_E_ARG;
  return ((int64_t)small_tiles(n) * (d->width[d->n_hidden - 1] + 1 + 2) + 16) * (int64_t)sizeof(float);
}
int dib_mlp_small_head_step(const dib_mlp_desc* d, const float* params, const float* x, int n, const float* y, int64_t ldy,
                            int loss_kind, float inv_global_batch, float* const* h, float* const* g, float* pred, float* g_pred,
                            float* g_x, float* grads, float* sums3, void* ws, dib_stream_t stream) {
  if (!d || !params || !x || !y || !h || !g || !pred || !g_pred || !grads || !sums3 || !ws || n <= 0) return DIB_E_ARG;
  if (loss_kind != DIB_LOSS_BCE_LOGITS && loss_kind != DIB_LOSS_MSE) return DIB_E_UNSUPPORTED;
  if (!dib_mlp_small_head_supported(d, n)) return DIB_E_UNSUPPORTED;
  DibSmallIntArgs a;
  std::memset(&a, 0, sizeof(a));
  a.mode = DIB_SMALL_INT_FWD | DIB_SMALL_INT_HEAD | DIB_SMALL_INT_HEAD_GRAD | DIB_SMALL_INT_BWD | DIB_SMALL_INT_HEAD_REDUCE |
           (g_x ? 0 : DIB_SMALL_INT_NO_GU);
  a.U = x; a.GU = g_x; a.batch = n; a.K0 = d->in_dim; a.params = params; a.n_hidden = d->n_hidden;
  for (int i = 0; i <= d->n_hidden; ++i) { a.width[i] = d->width[i]; a.w_off[i] = d->w_off[i]; a.b_off[i] = d->b_off[i]; }
  for (int i = 0; i < d->n_hidden; ++i) {
    if (!h[i] || !g[i]) return DIB_E_ARG;
    a.h[i] = h[i]; a.g[i] = g[i];
  }
  a.act = d->act; a.out_act = 0; a.out_dim = 1;
  a.pred = pred; a.g_pred = g_pred; a.loss_kind = loss_kind; a.Y = y; a.ldy = ldy; a.row_idx = nullptr; a.row0 = 0;
  a.inv_bg = inv_global_batch;
  const int tiles = small_tiles(n), KL = d->width[d->n_hidden - 1];
  float* w = (float*)ws;
  a.partial_w = w; a.partial_l = w + (int64_t)tiles * (KL + 1);
  a.sync = (unsigned*)(a.partial_l + 2 * tiles);   // zero at first use (the caller zero-fills the workspace once)
  a.head_gw = grads + d->w_off[d->n_hidden]; a.head_gb = grads + d->b_off[d->n_hidden];
  a.sums3 = sums3; a.loss_scale = inv_global_batch;
  const size_t lds = (size_t)mlp_head_lds_floats(d) * 4;
  static int lds_have[64] = {};
  if (int rc = ensure_dynamic_lds((const void*)dib_small_integration_kernel, lds, lds_have)) return rc;
  ProfScope ps(kProfOther, (hipStream_t)stream);
  DIB_LAUNCH(dib_small_integration_kernel, dim3(tiles), dim3(DIB_SMALL_THREADS), lds, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

// the companion protocol: arm, run the model's entry point, launch alone if the model's path had no row-tile launch to share
static int with_companion(const dib_mlp_desc* d, const DibSmallIntArgs& c, hipStream_t st, int model_rc_fn(void*), void* ctx) {
  t_companion.args = c;
  t_companion.lds = (size_t)mlp_small_lds_floats(d) * 4;
  t_companion.armed = true;
  int rc = model_rc_fn(ctx);
  if (t_companion.armed) {
    t_companion.armed = false;
    if (!rc) rc = mlp_small_launch(d, c, st);
  }
  return rc;
}
int dib_integration_fwd_and_mlp_fwd(dib_layout* l, int batch, const float* params, void* ws, const dib_mlp_desc* d,
                                    const float* mlp_params, const float* x, int64_t ldx, const int32_t* row_idx, int n, float* a0,
                                    float* const* h, float* out, dib_stream_t stream) {
  DibSmallIntArgs c;
  if (int rc = mlp_small_fwd_args(d, mlp_params, x, ldx, row_idx, n, a0, h, out, c)) return rc;
  struct Ctx { dib_layout* l; int batch; const float* params; void* ws; dib_stream_t stream; } ctx{l, batch, params, ws, stream};
  return with_companion(d, c, (hipStream_t)stream, [](void* p) {
    Ctx* q = (Ctx*)p;
    return dib_integration_fwd(q->l, q->batch, q->params, q->ws, q->stream);
  }, &ctx);
}
int dib_backward_and_mlp_bwd(dib_layout* l, int batch, const float* params, float* grads, const float* beta_dev,
                             float inv_global_batch, int flags, void* ws, const dib_mlp_desc* d, const float* mlp_params,
                             const float* g_out, float* const* h, float* const* g, int n, dib_stream_t stream) {
  DibSmallIntArgs c;
  if (int rc = mlp_small_bwd_args(d, mlp_params, g_out, h, g, n, c)) return rc;
  struct Ctx { dib_layout* l; int batch; const float* params; float* grads; const float* beta_dev; float inv; int flags; void* ws;
               dib_stream_t stream; } ctx{l, batch, params, grads, beta_dev, inv_global_batch, flags, ws, stream};
  return with_companion(d, c, (hipStream_t)stream, [](void* p) {
    Ctx* q = (Ctx*)p;
    return dib_backward(q->l, q->batch, q->params, q->grads, q->beta_dev, q->inv, q->flags, q->ws, q->stream);
  }, &ctx);
}

// grads = sum of the nsplit partial slabs (nsplit == 0: grads as given), Keras-Adam on (params, m, v), step count bumped - ONE
// launch (the generic segment of dib_step_tail_kernel) for parameter buffers that are not a dib_layout (dense.DenseStack)
static_assert(DIB_SYNC_WORDS == DIB_TAIL_SYNC_WORDS, "include/dib_hip.h DIB_SYNC_WORDS must cover the tail's arrival counters");
int dib_reduce_adam_step(const float* partial, int nsplit, int64_t stride, float* params, float* grads, float* adam_m,
                         float* adam_v, int64_t n, const float* lr_dev, int64_t* t_dev, float beta1, float beta2, float eps,
                         float grad_scale, uint32_t* sync, dib_stream_t stream) {
  if (!params || !grads || !adam_m || !adam_v || !lr_dev || !t_dev || !sync || n <= 0 || (n & 3) || nsplit < 0) return DIB_E_ARG;
  if (nsplit > 0 && (!partial || stride < n)) return DIB_E_ARG;
  DibTailArgs a;
  std::memset(&a, 0, sizeof(a));
  a.params = params; a.grads = grads; a.m = adam_m; a.v = adam_v; a.lr_dev = lr_dev; a.t_dev = (long long*)t_dev;
  a.b1 = beta1; a.b2 = beta2; a.eps = eps; a.gscale = grad_scale; a.flags = DIB_TAIL_ADAM | DIB_TAIL_BUMP;
  a.gbeg = 0; a.gend = n; a.slabs = partial; a.nsplit = nsplit; a.slab_stride = stride;
  a.nb_generic = (int)std::min<int64_t>(2048, (n / 4 + 255) / 256);
  a.sync = sync;
  ProfScope ps(kProfOther, (hipStream_t)stream);
  DIB_LAUNCH(dib_step_tail_kernel, dim3(a.nb_generic), dim3(256), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

int64_t dib_mi_workspace_bytes(int n, int E) {
  if (n <= 0 || E <= 0) return DIB_E_ARG;
  return (int64_t)sizeof(double) * (4ll * n * E + n);   // 1/sigma, u [N][E]; c [N]; mu, 1/sigma dimension-major [E][N]
}

int dib_mi_sandwich_rows(const float* enc_out, int n, int E, uint64_t seed, uint32_t step, uint32_t feature,
                         double* lower_rows, double* upper_rows, void* ws, dib_stream_t stream) {
  if (!enc_out || !lower_rows || !upper_rows || !ws || n <= 1 || E <= 0) return DIB_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  double* inv_sigma = (double*)ws;
  double* u = inv_sigma + (int64_t)n * E;
  double* cj = u + (int64_t)n * E;
  double* mu_t = cj + n;
  double* is_t = mu_t + (int64_t)n * E;
  DIB_LAUNCH(dib_mi_prep_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, enc_out, n, E, (unsigned long long)seed,
                     (unsigned)step, (unsigned)feature, inv_sigma, u, cj, mu_t, is_t);
  int rc = (int)hipGetLastError();
  if (rc) return rc;
  DIB_LAUNCH(dib_mi_rows_kernel, dim3(n), dim3(256), 0, st, enc_out, n, E, (const double*)inv_sigma,
                     (const double*)u, (const double*)cj, (const double*)mu_t, (const double*)is_t, lower_rows, upper_rows);
  return (int)hipGetLastError();
}

int dib_philox_normal_fill(float* eps, const int32_t* row_idx, int64_t row0, int batch, int F, int E, uint64_t seed,
                           uint32_t step, dib_stream_t stream) {
  if (!eps || batch <= 0 || F <= 0 || E <= 0) return DIB_E_ARG;
  DIB_LAUNCH(dib_eps_fill_kernel, dim3(grid_for((int64_t)batch * F * ((E + 3) / 4))), dim3(256), 0,
                     (hipStream_t)stream, eps, (const int*)row_idx, (long long)row0, batch, F, E,
                     (unsigned long long)seed, (unsigned)step);
  return (int)hipGetLastError();
}

int64_t dib_launch_count(void) { return (int64_t)g_dib_launches.load(std::memory_order_relaxed); }

int dib_profile_enable(int on) {
  std::lock_guard<std::mutex> lk(g_prof.mu);
  for (int c = 0; c < kProfCats; ++c) {
    for (auto& sp : g_prof.spans[c]) { g_prof.pool.push_back(sp.first); g_prof.pool.push_back(sp.second); }
    g_prof.spans[c].clear();
  }
  g_prof.on = on != 0;
  return DIB_OK;
}

int dib_profile_summary(double* ms_by_category, int* launches_by_category) {
  if (!ms_by_category || !launches_by_category) return DIB_E_ARG;
  std::lock_guard<std::mutex> lk(g_prof.mu);
  for (int c = 0; c < kProfCats; ++c) {
    double tot = 0.0;
    for (auto& sp : g_prof.spans[c]) {
      hipError_t e = hipEventSynchronize(sp.second);
      if (e != hipSuccess) return (int)e;
      float ms = 0.f;
      e = hipEventElapsedTime(&ms, sp.first, sp.second);
      if (e != hipSuccess) return (int)e;
      tot += ms;
    }
    ms_by_category[c] = tot;
    launches_by_category[c] = (int)g_prof.spans[c].size();
  }
  return DIB_OK;
}

float dib_philox_normal_ref(uint64_t seed, uint32_t step, uint32_t row, uint32_t feature, uint32_t e) {
  float out[4];
  dib_eps4(seed, step, row, feature, e >> 2, out);
  return out[e & 3];
}

int dib_gemm(int mode, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
             const float* bias, const float* aux, int ldaux, int act, void* dev_desc, dib_stream_t stream) {
  if (!A || !B || !C || !dev_desc || M <= 0 || N <= 0 || K <= 0 || mode < 0 || mode > 2) return DIB_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  DibGemmGroup g = make_group(Off(), lda, Off(), ldb, Off(), ldc, bias ? 0 : -1, Off(), ldaux, M, N, K);
  if ((((uintptr_t)A | (uintptr_t)B) & 15) != 0) return DIB_E_ARG;  // operands must be 16-byte aligned
  // descriptor travels BY VALUE in a kernel argument and is written on the stream (capture-safe: no host-memory copy node
  // pointing at this stack frame)
  DIB_LAUNCH(dib_write_desc_kernel, dim3(1), dim3(1), 0, st, (DibGemmGroup*)dev_desc, g);
  if (hipError_t e = hipGetLastError(); e != hipSuccess) return (int)e;
  const int tm = cdiv(M, 128), tn = cdiv(N, 128);
  const DibGemmGroup* dg = (const DibGemmGroup*)dev_desc;
  const dim3 g1(8 * cdiv(tm, 8) * tn, 1, 1);
  if (mode == 0)
    DIB_LAUNCH((dib_gemm_kernel<0, 2, 2, 32>), g1, dim3(256), 0, st, dg, A, B, C, bias, aux, (float*)nullptr, 0,
                       act, tm, tn, 0, 0ll);
  else if (mode == 1)
    DIB_LAUNCH((dib_gemm_kernel<1, 2, 2, 32>), g1, dim3(256), 0, st, dg, A, B, C, bias, aux, (float*)nullptr, 0,
                       act, tm, tn, 0, 0ll);
  else  // single split over the whole contraction; bias (if given) receives the column sums of B
    DIB_LAUNCH((dib_gemm_kernel<2, 2, 2, 32>), dim3(1, tm * tn, 1), dim3(256), 0, st, dg, A, B, C,
                       (const float*)nullptr, aux, (float*)bias, 0, act, tm, tn, K, 0ll);
  return (int)hipGetLastError();
}


// ---- set-transformer building blocks (include/dib_st.h) ----------------------------------------------------------------
static_assert(sizeof(dib_gemm_desc) == sizeof(DibGemmGroup), "public descriptor must mirror the kernel's group struct");

int dib_gemm_grouped(int mode, int n_groups, const dib_gemm_desc* dev_desc, int max_m, int max_n, const float* A,
                     const float* B, float* C, const float* bias, const float* aux, float* bias_out, int act, int nsplit,
                     int rows_per_split, int64_t split_stride, dib_stream_t stream) {
  if (!dev_desc || !A || !B || !C || n_groups <= 0 || max_m <= 0 || max_n <= 0 || mode < 0 || mode > 2 || !act_ok(act))
    return DIB_E_ARG;
  if (mode == 2 && (nsplit <= 0 || rows_per_split <= 0)) return DIB_E_ARG;
  GemmCall c;
  c.first = 0; c.count = n_groups; c.max_m = max_m; c.max_n = max_n;
  const DibGemmGroup* g = reinterpret_cast<const DibGemmGroup*>(dev_desc);
  hipStream_t st = (hipStream_t)stream;
  switch (mode) {
    case 0: return launch_gemm<0>(g, c, A, B, C, bias, aux, nullptr, 0, act, 1, 0, 0, st);
    case 1: return launch_gemm<1>(g, c, A, B, C, nullptr, aux, nullptr, 0, act, 1, 0, 0, st);
    default: return launch_gemm<2>(g, c, A, B, C, nullptr, nullptr, bias_out, 0, 0, nsplit, rows_per_split,
                                   (long long)split_stride, st);
  }
}

int dib_gemm_skinny_k(int mode, int n_groups, const dib_gemm_desc* dev_desc, int M, int N, int K, const float* A,
                      const float* B, float* C, const float* bias, dib_stream_t stream) {
  if (!dev_desc || !A || !B || !C || n_groups <= 0 || M <= 0 || N <= 0 || K <= 0 || mode < 0 || mode > 1) return DIB_E_ARG;
  if (K > 32 || (K & 3) || (N & 31) || n_groups > 65535 || cdiv(N, 128) > 65535) return DIB_E_UNSUPPORTED;
  const DibGemmGroup* g = reinterpret_cast<const DibGemmGroup*>(dev_desc);
  // >= ~4096 workgroups of 4 independent waves (16 wave slots per CU), at most 8 row tiles of 64 per workgroup
  const int total_tiles = cdiv(M, 64), cn = cdiv(N, 128);
  int chunks = std::max(1, std::min(total_tiles, cdiv(4096, cn * n_groups)));
  int tiles = std::min(8, cdiv(total_tiles, chunks));
  chunks = cdiv(total_tiles, tiles);
  const int nt_store = (long long)M * N * (long long)sizeof(float) * n_groups >= (256ll << 20) ? 1 : 0;
  const dim3 grid(chunks, cn, n_groups);
  hipStream_t st = (hipStream_t)stream;
  if (mode == 0)
    DIB_LAUNCH(dib_gemm_skinnyk_kernel<0>, grid, dim3(256), 0, st, g, A, B, C, bias, M, N, K, tiles, nt_store);
  else
    DIB_LAUNCH(dib_gemm_skinnyk_kernel<1>, grid, dim3(256), 0, st, g, A, B, C, (const float*)nullptr, M, N, K, tiles,
                       nt_store);
  return (int)hipGetLastError();
}

int dib_softmax_rows_fwd(float* S, int64_t rows, int P, int ld, float scale, dib_stream_t stream) {
  if (!S || rows <= 0 || P <= 0 || ld < P) return DIB_E_ARG;
  const dim3 grid(grid_for(rows, 4, 8192));
  hipStream_t st = (hipStream_t)stream;
#define DIB_SM(R) DIB_LAUNCH(dib_softmax_rows_fwd_kernel<R>, grid, dim3(256), 0, st, S, (long long)rows, P, ld, scale)
  if (P <= 64) DIB_SM(1); else if (P <= 256) DIB_SM(4); else if (P <= 1024) DIB_SM(16); else if (P <= 4096) DIB_SM(64); else DIB_SM(0);
#undef DIB_SM
  return (int)hipGetLastError();
}

int dib_softmax_rows_bwd(const float* Pm, float* dP, int64_t rows, int P, int ld, float scale, dib_stream_t stream) {
  if (!Pm || !dP || rows <= 0 || P <= 0 || ld < P) return DIB_E_ARG;
  const dim3 grid(grid_for(rows, 4, 8192));
  hipStream_t st = (hipStream_t)stream;
#define DIB_SM(R) DIB_LAUNCH(dib_softmax_rows_bwd_kernel<R>, grid, dim3(256), 0, st, Pm, dP, (long long)rows, P, ld, scale)
  if (P <= 64) DIB_SM(1); else if (P <= 256) DIB_SM(4); else if (P <= 1024) DIB_SM(16); else if (P <= 4096) DIB_SM(64); else DIB_SM(0);
#undef DIB_SM
  return (int)hipGetLastError();
}

static int ln_grid(int64_t T, int D) { return grid_for(T, D <= 32 ? 8 : 4, 512); }

int dib_add_layernorm_fwd(const float* a, const float* b, int b_slabs, int64_t b_stride, int64_t T, int D, const float* gamma,
                          const float* beta, float eps, float* y, float* xhat, float* rstd, dib_stream_t stream) {
  if (!a || !b || !gamma || !beta || !y || !xhat || !rstd || T <= 0 || D <= 0 || b_slabs < 1) return DIB_E_ARG;
  if (D > 256) return DIB_E_UNSUPPORTED;
  if (D <= 32)
    DIB_LAUNCH(dib_add_layernorm_fwd_kernel<32>, dim3(ln_grid(T, D)), dim3(256), 0, (hipStream_t)stream, a, b, b_slabs,
                       (long long)b_stride, (long long)T, D, gamma, beta, eps, y, xhat, rstd);
  else
    DIB_LAUNCH(dib_add_layernorm_fwd_kernel<64>, dim3(ln_grid(T, D)), dim3(256), 0, (hipStream_t)stream, a, b, b_slabs,
                       (long long)b_stride, (long long)T, D, gamma, beta, eps, y, xhat, rstd);
  return (int)hipGetLastError();
}

int64_t dib_add_layernorm_bwd_workspace_bytes(int64_t T, int D) {
  if (T <= 0 || D <= 0 || D > 256) return DIB_E_ARG;
  return (int64_t)ln_grid(T, D) * 4 * (D <= 32 ? 2 : 1) * 2 * D * (int64_t)sizeof(float);
}

int dib_add_layernorm_bwd_fused(const float* dy, const float* dy2, const float* xhat, const float* rstd, const float* gamma,
                                int64_t T, int D, float* ds, const float* act_src, int act, float* dz, float* dgamma_dbeta,
                                void* ws, dib_stream_t stream) {
  if (!dy || !xhat || !rstd || !gamma || !ds || !dgamma_dbeta || !ws || T <= 0 || D <= 0) return DIB_E_ARG;
  if ((dz != nullptr) != (act_src != nullptr) || (dz && !act_ok(act))) return DIB_E_ARG;
  if (D > 256) return DIB_E_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const int grid = ln_grid(T, D);
  float* partial = (float*)ws;
  if (D <= 32)
    DIB_LAUNCH(dib_add_layernorm_bwd_kernel<32>, dim3(grid), dim3(256), 0, st, dy, dy2, xhat, rstd, gamma, (long long)T, D,
                       ds, act_src, act, dz, partial);
  else
    DIB_LAUNCH(dib_add_layernorm_bwd_kernel<64>, dim3(grid), dim3(256), 0, st, dy, dy2, xhat, rstd, gamma, (long long)T, D,
                       ds, act_src, act, dz, partial);
  int rc = (int)hipGetLastError();
  if (rc) return rc;
  // [gamma gradient (D) | beta gradient (D)] = fixed-order column sums of the per-slot partials
  DIB_LAUNCH(dib_colsum_partials_kernel, dim3(2 * D), dim3(256), 0, st, (const float*)partial,
                     grid * 4 * (D <= 32 ? 2 : 1), 2 * D, dgamma_dbeta);
  return (int)hipGetLastError();
}

int dib_add_layernorm_bwd(const float* dy, const float* xhat, const float* rstd, const float* gamma, int64_t T, int D,
                          float* ds, float* dgamma_dbeta, void* ws, dib_stream_t stream) {
  return dib_add_layernorm_bwd_fused(dy, nullptr, xhat, rstd, gamma, T, D, ds, nullptr, 0, nullptr, dgamma_dbeta, ws, stream);
}

// ---- the token-wise half of a set-transformer block in one launch per direction (csrc/dib_st_chain.h) ----------------
static_assert(sizeof(dib_st_block_desc) == sizeof(DibStChainDesc), "public block descriptor must mirror the kernel's");
static size_t st_chain_fwd_lds(const dib_st_block_desc* d) {
  size_t fl = (size_t)DIB_SMALL_ROWS * (dib_small_pitch(d->HK) + 2 * dib_small_pitch(d->D)) + DIB_SMALL_XCH_FLOATS;
  for (int l = 0; l < d->n_ff; ++l) fl += (size_t)DIB_SMALL_ROWS * dib_small_pitch(d->ff_width[l]);
  return fl * sizeof(float);
}
static size_t st_chain_bwd_lds(const dib_st_block_desc* d) {
  size_t fl = (size_t)DIB_SMALL_ROWS * (4 * dib_small_pitch(d->D) + 2 * d->D) + DIB_SMALL_XCH_FLOATS;
  for (int l = 0; l < d->n_ff; ++l) fl += 2 * (size_t)DIB_SMALL_ROWS * dib_small_pitch(d->ff_width[l]);
  return fl * sizeof(float);
}

int dib_st_chain_supported(const dib_st_block_desc* d, int64_t T) {
  if (!d || T <= 0 || !knobs().small_batch) return 0;
  if (d->D <= 0 || d->D % 32 || d->D > 256 || d->HK <= 0 || d->HK % 16 || d->n_ff < 1 || d->n_ff > DIB_ST_CHAIN_MAX_FF) return 0;
  if (d->act < 0 || d->act > 2) return 0;
  for (int l = 0; l < d->n_ff; ++l)
    if (d->ff_width[l] <= 0 || d->ff_width[l] % 16 || d->ff_width[l] > 1024) return 0;
  if (d->ff_width[d->n_ff - 1] != d->D) return 0;
  if (T > 4096) return 0;   // above: one tiled GEMM per layer reads each weight once per 64-128 rows instead of once per 16
  return st_chain_fwd_lds(d) <= 150 * 1024 && st_chain_bwd_lds(d) <= 150 * 1024;
}

int64_t dib_st_chain_workspace_bytes(int64_t T, int D) {
  if (T <= 0 || D <= 0) return DIB_E_ARG;
  return ((T + DIB_SMALL_ROWS - 1) / DIB_SMALL_ROWS * 4 * D + 64) * (int64_t)sizeof(float);
}

int dib_st_chain_fwd(const dib_st_block_desc* d, int64_t T, const float* params, const float* ctx, const float* x_in, float* h,
                     float* xhat1, float* rstd1, float* const* ff, float* x_out, float* xhat2, float* rstd2, dib_stream_t stream) {
  if (!d || !params || !ctx || !x_in || !h || !xhat1 || !rstd1 || !ff || !x_out || !xhat2 || !rstd2) return DIB_E_ARG;
  if (!dib_st_chain_supported(d, T)) return DIB_E_UNSUPPORTED;
  DibStChainFwdArgs a;
  std::memset(&a, 0, sizeof(a));
  std::memcpy(&a.d, d, sizeof(a.d));
  a.T = T; a.params = params; a.ctx = ctx; a.x_in = x_in; a.h = h; a.xhat1 = xhat1; a.rstd1 = rstd1;
  for (int l = 0; l < d->n_ff; ++l) { if (!ff[l]) return DIB_E_ARG; a.ff[l] = ff[l]; }
  a.x_out = x_out; a.xhat2 = xhat2; a.rstd2 = rstd2;
  const size_t lds = st_chain_fwd_lds(d);
  static int lds_have[64] = {};
  if (int rc = ensure_dynamic_lds((const void*)dib_st_chain_fwd_kernel, lds, lds_have)) return rc;
  DIB_LAUNCH(dib_st_chain_fwd_kernel, dim3((unsigned)((T + DIB_SMALL_ROWS - 1) / DIB_SMALL_ROWS)), dim3(DIB_SMALL_THREADS), lds,
             (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

int dib_st_chain_bwd(const dib_st_block_desc* d, int64_t T, const float* params, const float* g_out, int g_out_slabs,
                     int64_t g_out_stride, const float* xhat2, const float* rstd2, const float* const* ff, const float* xhat1,
                     const float* rstd1, float* const* g_ff, float* g_in, float* g_ctx, float* grads, void* ws, dib_stream_t stream) {
  if (!d || !params || !g_out || !xhat2 || !rstd2 || !ff || !xhat1 || !rstd1 || !g_ff || !g_in || !g_ctx || !grads || !ws)
    return DIB_E_ARG;
  if (g_out_slabs < 1 || (g_out_slabs > 1 && (g_out_stride < T * d->D || (g_out_stride & 3))) || ((uintptr_t)g_out & 15)) return DIB_E_ARG;
  if (!dib_st_chain_supported(d, T)) return DIB_E_UNSUPPORTED;
  DibStChainBwdArgs a;
  std::memset(&a, 0, sizeof(a));
  std::memcpy(&a.d, d, sizeof(a.d));
  a.T = T; a.params = params; a.g_out = g_out; a.g_slabs = g_out_slabs; a.g_stride = g_out_stride;
  a.xhat2 = xhat2; a.rstd2 = rstd2; a.xhat1 = xhat1; a.rstd1 = rstd1;
  for (int l = 0; l < d->n_ff; ++l) { if (!ff[l] || !g_ff[l]) return DIB_E_ARG; a.ff[l] = ff[l]; a.g_ff[l] = g_ff[l]; }
  a.g_in = g_in; a.g_ctx = g_ctx; a.grads = grads;
  const long long tiles = (T + DIB_SMALL_ROWS - 1) / DIB_SMALL_ROWS;
  a.ln_partial = (float*)ws;
  a.sync = (unsigned*)((float*)ws + tiles * 4 * d->D);   // zero at first use (the caller zero-fills the workspace once)
  const size_t lds = st_chain_bwd_lds(d);
  static int lds_have[64] = {};
  if (int rc = ensure_dynamic_lds((const void*)dib_st_chain_bwd_kernel, lds, lds_have)) return rc;
  DIB_LAUNCH(dib_st_chain_bwd_kernel, dim3((unsigned)tiles), dim3(DIB_SMALL_THREADS), lds, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

int dib_mean_pool_fwd(const float* x, int B, int P, int D, float* out, dib_stream_t stream) {
  if (!x || !out || B <= 0 || P <= 0 || D <= 0) return DIB_E_ARG;
  DIB_LAUNCH(dib_mean_pool_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, x, B, P, D, out);
  return (int)hipGetLastError();
}

int dib_mean_pool_bwd(const float* g, int B, int P, int D, float* dx, dib_stream_t stream) {
  if (!g || !dx || B <= 0 || P <= 0 || D <= 0) return DIB_E_ARG;
  DIB_LAUNCH(dib_mean_pool_bwd_kernel, dim3(grid_for((int64_t)B * P * D)), dim3(256), 0, (hipStream_t)stream, g, B, P,
                     D, dx);
  return (int)hipGetLastError();
}

int dib_add_inplace(float* dst, const float* src, int64_t n, dib_stream_t stream) {
  if (!dst || !src || n <= 0) return DIB_E_ARG;
  DIB_LAUNCH(dib_add_inplace_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, dst, src, (long long)n);
  return (int)hipGetLastError();
}

int64_t dib_attention_stash_bytes(int B, int P, int H) {
  if (B <= 0 || P <= 0 || H <= 0) return DIB_E_ARG;
  if (P <= kAttnSmallP) return 0;   // the single-workgroup path keeps the scores in LDS: nothing to stash
  const int64_t nt = cdiv(P, kAttnTile);
  return (int64_t)sizeof(float) * B * H * nt * nt * kAttnTile * kAttnTile;
}

int dib_attention_fwd(const float* q, const float* k, const float* v, int B, int P, int H, int key_dim, int64_t ld,
                      float scale, float* o, float* lse, float* s_stash, dib_stream_t stream) {
  if (!q || !k || !v || !o || !lse || B <= 0 || P <= 0 || H <= 0 || ld < (int64_t)H * key_dim || (ld & 3)) return DIB_E_ARG;
  if (key_dim != kAttnD || (int64_t)P * ld >= (1ll << 30)) return DIB_E_UNSUPPORTED;   // 32-bit row offsets inside one neighbourhood
  if ((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o | (uintptr_t)s_stash) & 15) != 0) return DIB_E_ARG;
  DibAttnArgs a{};
  a.q = q; a.k = k; a.v = v; a.o = o; a.lse = lse; a.s_stash = s_stash; a.P = P; a.H = H; a.ld = ld; a.scale = scale;
  ProfScope ps(kProfAttnFwd, (hipStream_t)stream);
  if (P <= kAttnSmallP) {   // the whole head in LDS, one workgroup per (neighbourhood, head): csrc/dib_attn_small.h (no stash)
    const size_t lds = (size_t)DibAttnSmallFwdLds * sizeof(float);
    static std::atomic<bool> attr_small[64];
    if (AttrOnce once(attr_small); once) {
      hipError_t e = hipFuncSetAttribute((const void*)dib_attn_small_fwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return (int)e;
    }
    DIB_LAUNCH(dib_attn_small_fwd_kernel<false>, dim3(H, B), dim3(256), lds, (hipStream_t)stream, a);
    return (int)hipGetLastError();
  }
  if (knobs().attn_fwd_waves == 8 && P >= 256) DIB_LAUNCH(dib_attn_fwd8_kernel, dim3(cdiv(P, 256), H, B), dim3(512), 0, (hipStream_t)stream, a);
  else DIB_LAUNCH(dib_attn_fwd_kernel, dim3(cdiv(P, 128), H, B), dim3(256), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

int dib_attention_fwd_proj_supported(int P, int key_dim, int model_dim) {
  return P >= 1 && P <= kAttnSmallP && key_dim == kAttnD && model_dim == 32;
}

int dib_attention_fwd_proj(const float* x, int64_t ldx, const float* params, const int64_t* w_off, const int64_t* b_off, int B, int P,
                           int H, int key_dim, int model_dim, int64_t ld, float scale, float* q, float* k, float* v, float* o,
                           float* lse, dib_stream_t stream) {
  if (!x || !params || !w_off || !b_off || !q || !k || !v || !o || !lse || B <= 0 || P <= 0 || H <= 0 || ldx < model_dim || (ldx & 3))
    return DIB_E_ARG;
  if (!dib_attention_fwd_proj_supported(P, key_dim, model_dim)) return DIB_E_UNSUPPORTED;
  if (ld != (int64_t)H * key_dim) return DIB_E_ARG;   // the projection kernels [model_dim][H * key_dim] share the outputs' leading dimension
  if ((((uintptr_t)x | (uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o) & 15) != 0) return DIB_E_ARG;
  DibAttnArgs a{};
  a.o = o; a.lse = lse; a.P = P; a.H = H; a.ld = ld; a.scale = scale;
  a.px = x; a.pldx = ldx; a.pparams = params; a.pq = q; a.pk = k; a.pv = v;
  for (int i = 0; i < 3; ++i) { a.pw[i] = w_off[i]; a.pb[i] = b_off[i]; }
  ProfScope ps(kProfAttnFwd, (hipStream_t)stream);
  const size_t lds = (size_t)DibAttnSmallFwdLds * sizeof(float);
  static std::atomic<bool> attr_small[64];
  if (AttrOnce once(attr_small); once) {
    hipError_t e = hipFuncSetAttribute((const void*)dib_attn_small_fwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  DIB_LAUNCH(dib_attn_small_fwd_kernel<true>, dim3(H, B), dim3(256), lds, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

int dib_attention_bwd_proj(const float* q, const float* k, const float* v, const float* d_o, const float* lse, int B, int P, int H,
                           int key_dim, int model_dim, int64_t ld, float scale, float* dq, float* dk, float* dv, const float* params,
                           const int64_t* w_off, float* dx_slabs, int64_t slab_stride, dib_stream_t stream) {
  if (!q || !k || !v || !d_o || !lse || !dq || !dk || !dv || !params || !w_off || !dx_slabs || B <= 0 || P <= 0 || H <= 0)
    return DIB_E_ARG;
  if (!dib_attention_fwd_proj_supported(P, key_dim, model_dim)) return DIB_E_UNSUPPORTED;
  if (ld != (int64_t)H * key_dim || slab_stride < (int64_t)B * P * model_dim || (slab_stride & 3)) return DIB_E_ARG;
  if ((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)d_o | (uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv |
        (uintptr_t)dx_slabs) & 15) != 0)
    return DIB_E_ARG;
  DibAttnArgs a{};
  a.q = q; a.k = k; a.v = v; a.lse = const_cast<float*>(lse); a.d_o = d_o; a.dq = dq; a.dk = dk; a.dv = dv;
  a.P = P; a.H = H; a.ld = ld; a.scale = scale;
  a.pparams = params; a.pdx = dx_slabs; a.pdx_stride = slab_stride;
  for (int i = 0; i < 3; ++i) { if (w_off[i] & 3) return DIB_E_ARG; a.pw[i] = w_off[i]; }
  const size_t lds = (size_t)DibAttnSmallBwdLds * sizeof(float);
  static std::atomic<bool> attr_small[64];
  if (AttrOnce once(attr_small); once) {
    hipError_t e = hipFuncSetAttribute((const void*)dib_attn_small_bwd8_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  ProfScope ps(kProfAttnBwd, (hipStream_t)stream);
  DIB_LAUNCH(dib_attn_small_bwd8_kernel<true>, dim3(H, B), dim3(512), lds, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

int64_t dib_attention_bwd_workspace_bytes(int B, int P, int H) {
  if (B <= 0 || P <= 0 || H <= 0) return DIB_E_ARG;
  const int64_t nkb = cdiv(P, 128);
  return (int64_t)sizeof(float) * ((int64_t)B * H * P + (nkb > 1 ? (int64_t)B * H * nkb * P * kAttnD : 0) + 64);
}

int dib_attention_bwd(const float* q, const float* k, const float* v, const float* o, const float* d_o, const float* lse,
                      const float* s_stash, int B, int P, int H, int key_dim, int64_t ld, float scale, float* dq, float* dk,
                      float* dv, void* ws, dib_stream_t stream) {
  if (!q || !k || !v || !o || !d_o || !lse || !dq || !dk || !dv || !ws || B <= 0 || P <= 0 || H <= 0 ||
      ld < (int64_t)H * key_dim || (ld & 3))
    return DIB_E_ARG;
  if (key_dim != kAttnD || (int64_t)P * ld >= (1ll << 30)) return DIB_E_UNSUPPORTED;   // 32-bit row offsets inside one neighbourhood
  if ((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o | (uintptr_t)d_o | (uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv |
        (uintptr_t)ws | (uintptr_t)s_stash) & 15) != 0)
    return DIB_E_ARG;   // every one of them is accessed with 16-byte loads / stores
  hipStream_t st = (hipStream_t)stream;
  if (P <= kAttnSmallP) {   // csrc/dib_attn_small.h: one launch - delta, the score recompute and dQ/dK/dV inside one workgroup per head
    DibAttnArgs a{};
    a.q = q; a.k = k; a.v = v; a.lse = const_cast<float*>(lse); a.d_o = d_o; a.dq = dq; a.dk = dk; a.dv = dv;
    a.P = P; a.H = H; a.ld = ld; a.scale = scale;
    const size_t lds = (size_t)DibAttnSmallBwdLds * sizeof(float);
    static std::atomic<bool> attr_small[64];
    if (AttrOnce once(attr_small); once) {
      hipError_t e = hipFuncSetAttribute((const void*)dib_attn_small_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)dib_attn_small_bwd8_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return (int)e;
    }
    ProfScope ps(kProfAttnBwd, st);
    if (knobs().attn_small_bwd_waves >= 8) DIB_LAUNCH(dib_attn_small_bwd8_kernel<false>, dim3(H, B), dim3(512), lds, st, a);
    else DIB_LAUNCH(dib_attn_small_bwd_kernel, dim3(H, B), dim3(256), lds, st, a);
    return (int)hipGetLastError();
  }
  float* delta = (float*)ws;
  float* part = delta + (((int64_t)B * H * P + 63) / 64) * 64;
  const int nkb = cdiv(P, 128);
  DIB_LAUNCH(dib_attn_delta_kernel, dim3(cdiv((int64_t)B * P * H, 4)), dim3(256), 0, st, o, d_o, (long long)ld, B, P, H,
                     delta);
  DibAttnArgs a{};
  a.q = q; a.k = k; a.v = v; a.lse = const_cast<float*>(lse); a.d_o = d_o; a.delta = delta; a.dq = dq; a.dk = dk; a.dv = dv;
  a.s_stash = const_cast<float*>(s_stash);
  a.P = P; a.H = H; a.ld = ld; a.scale = scale;
  const size_t lds = (size_t)DibAttnBwdLds * sizeof(float);
  static std::atomic<bool> attr_set[64];
  if (AttrOnce once(attr_set); once) {
    hipError_t e = hipFuncSetAttribute((const void*)dib_attn_bwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess)
      e = hipFuncSetAttribute((const void*)dib_attn_bwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  { ProfScope ps(kProfAttnBwd, st);
    if (s_stash) DIB_LAUNCH(dib_attn_bwd_kernel<true>, dim3(nkb, H, B), dim3(256), lds, st, a, part, nkb);
    else DIB_LAUNCH(dib_attn_bwd_kernel<false>, dim3(nkb, H, B), dim3(256), lds, st, a, part, nkb); }
  int rc = (int)hipGetLastError();
  if (rc) return rc;
  if (nkb > 1) {
    DIB_LAUNCH(dib_attn_dq_reduce_kernel, dim3(grid_for((int64_t)B * H * P * (kAttnD / 4))), dim3(256), 0, st,
                       (const float*)part, B, P, H, nkb, (long long)ld, scale, dq);
    rc = (int)hipGetLastError();
  }
  return rc;
}

#ifdef DIB_FUSED_TIMING
// diagnostic build only (not declared in include/): phase timers of the last fused encoder forward
extern "C" int dib_fused_debug_read(long long* out16) {
  if (hipDeviceSynchronize() != hipSuccess) return DIB_E_ARG;
  return (int)hipMemcpyFromSymbol(out16, HIP_SYMBOL(dib_fused_dbg), 16 * sizeof(long long));
}
#endif

#ifdef DIB_SMALL_TIMING
// diagnostic build only (not declared in include/): phase marks of the last launches of the row-tile kernels (dib_small.h)
extern "C" int dib_small_debug_read(long long* out64) {
  if (hipDeviceSynchronize() != hipSuccess) return DIB_E_ARG;
  return (int)hipMemcpyFromSymbol(out64, HIP_SYMBOL(dib_small_dbg), 64 * sizeof(long long));
}
#endif

#ifdef DIB_ATTN_TIMING
// diagnostic build only (not declared in include/): copy the phase timers of the last dib_attention_bwd to the host
extern "C" int dib_attn_debug_read(long long* out16) {
  if (hipDeviceSynchronize() != hipSuccess) return DIB_E_ARG;
  return (int)hipMemcpyFromSymbol(out16, HIP_SYMBOL(dib_attn_dbg), 16 * sizeof(long long));
}
#endif

int dib_act_grad_mul(const float* g, const float* y, int act, int64_t n, float* out, dib_stream_t stream) {
  if (!g || !y || !out || n <= 0 || !act_ok(act)) return DIB_E_ARG;
  DIB_LAUNCH(dib_act_grad_mul_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, g, y, act, (long long)n, out);
  return (int)hipGetLastError();
}

int64_t dib_token_kl_workspace_bytes(int64_t T, int E) {
  if (T <= 0 || T > 0x7fffffff || E <= 0 || (E + 3) / 4 > 256) return DIB_E_ARG;   // same limit as the fwd / bwd entries
  return (int64_t)cdiv(T, std::max(1, 256 / ((E + 3) / 4))) * (int64_t)sizeof(float);
}

int dib_token_reparam_kl_fwd(const float* enc_out, int64_t T, int E, float logvar_offset, uint64_t seed, uint32_t step,
                             const uint32_t* step_dev, int64_t row0, int deterministic, float* u, float* kl_sum, void* ws,
                             dib_stream_t stream) {
  if (!enc_out || !u || !kl_sum || !ws || T <= 0 || T > 0x7fffffff || E <= 0 || (E + 3) / 4 > 256) return DIB_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int blocks = cdiv(T, std::max(1, 256 / ((E + 3) / 4)));
  DIB_LAUNCH(dib_reparam_kl_fwd_kernel, dim3(blocks, 1), dim3(256), 0, st, enc_out, u, (float*)ws, (const int*)nullptr,
                     (long long)row0, (int)T, 1, E, (unsigned long long)seed, (unsigned)step, deterministic ? 1 : 0,
                     (const unsigned*)step_dev, logvar_offset);
  int rc = (int)hipGetLastError();
  if (rc) return rc;
  DIB_LAUNCH(dib_colsum_partials_kernel, dim3(1), dim3(256), 0, st, (const float*)ws, blocks, 1, kl_sum);
  return (int)hipGetLastError();
}

int dib_token_reparam_kl_bwd(const float* enc_out, const float* g_u, const float* u, int64_t T, int E, float logvar_offset,
                             const float* beta_dev, float inv_batch, float* d_enc_out, dib_stream_t stream) {
  if (!enc_out || !g_u || !u || !beta_dev || !d_enc_out || T <= 0 || T > 0x7fffffff || E <= 0 || (E + 3) / 4 > 256)
    return DIB_E_ARG;
  const int blocks = cdiv(T, std::max(1, 256 / ((E + 3) / 4)));
  DIB_LAUNCH(dib_reparam_kl_bwd_kernel, dim3(blocks, 1), dim3(256), 0, (hipStream_t)stream, enc_out, g_u, u, d_enc_out,
                     beta_dev, inv_batch, (int)T, 1, E, logvar_offset);
  return (int)hipGetLastError();
}

int64_t dib_mi_probe_workspace_bytes(int n_probes, int n_data, int E) {
  if (n_probes <= 0 || n_data <= 0 || E <= 0) return DIB_E_ARG;
  return (int64_t)sizeof(double) * ((4ll * E + 1) * ((int64_t)n_probes + n_data));   // per point set as dib_mi_workspace_bytes
}

int dib_mi_probe_bounds(const float* enc_probe, int n_probes, const float* enc_data, int n_data, int E, float logvar_offset,
                        uint64_t seed, uint32_t step, uint32_t feature, double* lower_rows, double* upper_rows,
                        double* u_probe_out, void* ws, dib_stream_t stream) {
  if (!enc_probe || !enc_data || !lower_rows || !upper_rows || !ws || n_probes <= 0 || n_data <= 0 || E <= 0) return DIB_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  double* is_p = (double*)ws;
  double* u_p = is_p + (int64_t)n_probes * E;
  double* c_p = u_p + (int64_t)n_probes * E;
  double* mut_p = c_p + n_probes;
  double* ist_p = mut_p + (int64_t)n_probes * E;
  double* is_d = ist_p + (int64_t)n_probes * E;
  double* u_d = is_d + (int64_t)n_data * E;
  double* c_d = u_d + (int64_t)n_data * E;
  double* mut_d = c_d + n_data;
  double* ist_d = mut_d + (int64_t)n_data * E;
  DIB_LAUNCH(dib_mi_prep_kernel, dim3(cdiv(n_probes, 256)), dim3(256), 0, st, enc_probe, n_probes, E,
                     (unsigned long long)seed, (unsigned)step, (unsigned)feature, is_p, u_p, c_p, mut_p, ist_p, logvar_offset);
  DIB_LAUNCH(dib_mi_prep_kernel, dim3(cdiv(n_data, 256)), dim3(256), 0, st, enc_data, n_data, E,
                     (unsigned long long)seed, (unsigned)step, (unsigned)feature + 1u, is_d, u_d, c_d, mut_d, ist_d, logvar_offset);
  int rc = (int)hipGetLastError();
  if (rc) return rc;
  DIB_LAUNCH(dib_mi_probe_rows_kernel, dim3(n_probes), dim3(256), 0, st, enc_probe, (const double*)u_p,
                     (const double*)is_p, (const double*)c_p, (const double*)mut_d, (const double*)ist_d, (const double*)c_d,
                     n_data, E, lower_rows, upper_rows);
  rc = (int)hipGetLastError();
  if (rc) return rc;
  if (u_probe_out)
    return (int)hipMemcpyAsync(u_probe_out, u_p, (size_t)n_probes * E * sizeof(double), hipMemcpyDeviceToDevice, st);
  return DIB_OK;
}

int64_t dib_loss_rows_workspace_bytes(int batch) {
  if (batch <= 0) return DIB_E_ARG;
  return (int64_t)cdiv(batch, 256) * 2 * (int64_t)sizeof(float);
}

int dib_loss_rows(int loss_kind, const float* pred, int out_dim, const float* y, int64_t ldy, int batch,
                  float inv_global_batch, float* g_pred, float* out3, void* ws, dib_stream_t stream) {
  if (!pred || !y || !g_pred || !out3 || !ws || batch <= 0 || out_dim <= 0) return DIB_E_ARG;
  if (loss_kind < 0 || loss_kind > 3) return DIB_E_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const int blocks = cdiv(batch, 256);
  DIB_LAUNCH(dib_loss_kernel, dim3(blocks), dim3(256), 0, st, loss_kind, pred, out_dim, y, (long long)ldy,
                     (const int*)nullptr, 0ll, batch, inv_global_batch, 0, g_pred, (float*)ws);
  int rc = (int)hipGetLastError();
  if (rc) return rc;
  DIB_LAUNCH(dib_loss_finalize_kernel, dim3(2), dim3(256), 0, st, (const float*)ws, blocks, (float)batch, out3);
  return (int)hipGetLastError();
}

int dib_reduce_splits(const float* partial, int64_t n, int nsplit, int64_t stride, float* out, dib_stream_t stream) {
  if (!partial || !out || n <= 0 || nsplit <= 0 || (n & 3) || (stride & 3)) return DIB_E_ARG;
  DIB_LAUNCH(dib_reduce_splits_kernel, dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, partial,
                     (long long)n, nsplit, (long long)stride, out, (const float*)nullptr);
  return (int)hipGetLastError();
}

int dib_reduce_splits_add(const float* partial, int64_t n, int nsplit, int64_t stride, float* out, dib_stream_t stream) {
  if (!partial || !out || n <= 0 || nsplit <= 0 || (n & 3) || (stride & 3)) return DIB_E_ARG;
  DIB_LAUNCH(dib_reduce_splits_kernel, dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, partial,
                     (long long)n, nsplit, (long long)stride, out, (const float*)out);
  return (int)hipGetLastError();
}

}  // extern "C"
