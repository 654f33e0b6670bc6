// Host-side schedules: which kernels run, in which order, on which buffers, for
//   UnetGenerator      (lib/models/networks.py:216-324)  forward + backward
//   PatchGANDiscriminator (networks.py:331-363)          forward + backward
// Data layout in HBM (per activation slot): NHWC tensors of the compute type T; for every U-Net
// level k one "concat buffer" C[k] = [skip a_k | decoder u_{k+1}] (2*ch_k channels) that both
// producers write at their channel offset, so torch.cat (networks.py:324) never materialises.
#include <string>
#include <vector>
#include <string.h>

#include "common.h"
#include "stat_acc.h"

namespace {

struct TensorDesc {
  std::string name;
  int kind;  // 0 conv weight, 1 contiguous param, 2 running_mean, 3 running_var
  int64_t shape[4];
  int ndim;
  int64_t offset, numel;
};

struct Conv {            // 4x4 weight [a][16][b] fp32 master at params+w_off
  int ca = 0, cb = 0;
  int64_t w_off = -1, bias_off = -1;
  int64_t packed_off = -1, phase_off = -1;  // workspace offsets of the T copies (-1: b == 1 layer)
  int64_t inf_off = -1;                     // inference: the forward's copy with the following BatchNorm's scale folded in
};
struct BN {
  int c = 0;
  int64_t gamma_off = -1, beta_off = -1, rmean_off = -1, rvar_off = -1;
  int64_t stat_off = -1;  // per-slot: [scale|shift|mean|inv] * c floats (offset inside slot)
  int id = -1;            // index into gi_net::eval_gen
  int64_t inf_off = -1;   // shared workspace: [scale|shift|mean|inv] * c floats of the inference affine map
  // exact-statistics accumulators (stat_acc.h), 64-bit words into the net's region: four blocks of [2 groups][c][4] words,
  // two for the forward statistics and two for the backward reductions. Each use takes the block its parity bit names
  // and its consuming pass clears the other one for the next use (ping-pong): no memset launches.
  int64_t acc_off = -1;
  int64_t in_off = -1;    // InstanceNorm generator: per-slot [n][c][2] mean / inv of the last forward
  mutable int fwd_par = 0, bwd_par = 0;
  mutable int dirty[4] = {0, 0, 0, 0};   // replicas a use has added to in block {fwd 0, fwd 1, bwd 0, bwd 1}, not yet cleared
  int64_t acc_block() const { return gi_stat_block_words(c, GI_STAT_MAXREP); }
};

// BatchNorm-backward reduction fused into the GEMM that produces the gradient (IgemmArgs::bwd_*): fixed BEFORE that GEMM
// runs (the accumulator ping-pong state moves here), consumed by act_bn_bwd, which then launches only its apply pass - or
// both passes on the same accumulator block when the kernel that ran did not take the fusion (`applied` false).
struct BwdFuse {
  bool planned = false, applied = false;
  unsigned long long* acc = nullptr; unsigned long long* zero_next = nullptr;
  int zero_words = 0, reps_gemm = 1, reps_reduce = 1, groups = 1;
  const void* x = nullptr; int ldx = 0;
  const float* scale = nullptr; const float* shift = nullptr; const float* mean = nullptr; const float* inv = nullptr;
  int stride = 0; float slope = 1.f; int64_t pg = 0;
  int c0 = 0, c = 0;   // IgemmArgs::bwd_c0 / bwd_c (0: all output columns)
};

struct Arena {
  int64_t size = 0;
  int64_t take(int64_t bytes) {
    const int64_t o = size;
    size += gi_align_up(bytes < 16 ? 16 : bytes, 256);
    return o;
  }
};

}  // namespace

struct gi_net {
  gi_ctx* ctx = nullptr;
  int kind = 0;  // 0 unet, 1 patchgan
  int dtype = GI_F32;
  int H = 0, W = 0, max_n = 0, n_slots = 1;
  int train = 1;
  float loss_scale = 1.f;
  uint64_t drop_seed = 0x5EED0000ull, drop_counter = 0;
  // unet
  int nd = 0, ngf = 0;
  int bn_groups = 1;                 // discriminator: the batch holds this many independent BatchNorm groups (stacked real|fake)
  int bwd_eval = 0;                  // (transient) the backward in progress differentiates an eval-mode forward
  int out_c = 1;                     // output channels; > 1: u1 runs on the generic kernels with b zero-padded to 64
  int64_t oU1 = -1, oUp1Pad = -1, oUp1Packed = -1, oUp1Phase = -1;
  float dropout_p = 0.f;
  std::vector<int> ch, Hk, Wk;       // 1-based per level
  std::vector<Conv> conv, up;        // 1-based
  std::vector<BN> dnorm, unorm;      // 1-based (c == 0: absent)
  // patchgan
  int sigmoid = 1;
  Conv dconv[5];                     // 1..4
  BN dbn[5];                         // 2..4
  int64_t w5_off = -1, wl_off = -1, bl_off = -1;
  int P = 0, Hh = 0, Wh = 0;

  std::vector<TensorDesc> tensors;
  int64_t n_params = 0, n_buffers = 0;
  // workspace layout
  Arena arena;
  int64_t slot_bytes = 0, slot_base = 0;
  std::vector<int64_t> oC, oR, oU, oMask;  // per-slot offsets (1-based level)
  int64_t oE = -1, oOut = -1, oX = -1, oStats = -1, oHh = -1;
  int64_t oA[5] = {-1, -1, -1, -1, -1}, oRd[5] = {-1, -1, -1, -1, -1};
  // shared scratch
  std::vector<int64_t> ogC, ogA;
  int64_t ogE = -1, oD = -1, oG0 = -1, oPart = -1, oSums = -1, oSplit = -1, oDh = -1, oCol = -1;
  int64_t part_floats = 0, split_bytes = 0;
  int64_t oHw = -1, hw_bytes = 0;    // head weight-gradient partials
  int64_t oHt = -1, ht_bytes = 0;    // head forward: per-pixel tap products of large feature maps (sliced forward)
  // eval-mode BatchNorm is a fixed affine map of (gamma, beta, running statistics): its scale / shift in a slot stay
  // valid until those change (weights re-synced, a train-mode forward, set_train), tracked by a generation counter
  uint64_t affine_gen = 1;
  uint64_t inf_gen = 0;              // generation the folded inference copies were built at
  int inference = 0;                 // gi_net_set_inference: eval-mode forwards will never be differentiated
  std::vector<int> slot_inference;
  std::vector<std::vector<uint64_t>> eval_gen;   // [slot][BN id]: generation the slot's scale / shift were computed at
  // generator norm layer (get_norm_layer, networks.py:29-45): 0 BatchNorm2d, 1 InstanceNorm2d(affine=False,
  // track_running_stats=False) - every convolution then carries a bias (use_bias, networks.py:270-273) -, 2 none (Identity)
  int norm_kind = 0;
  int64_t oInStats = -1;
  int fuse_head = 1;                 // generator: last decoder level's BatchNorm + ReLU applied inside the head kernels
  std::vector<int> slot_fused_u2;    // per slot: the forward ran that way (the backward must match)
  int64_t oWg = -1, wg_bytes = 0;    // weight-gradient split scratch (deterministic two-stage reduction)
  // GI_WGRAD_STREAM: the weight-gradient GEMMs of a backward (and their fixed-order reduce launches) run on a second HIP stream
  // beside the input-gradient chain, which never reads their result. dz buffers rotate (oDr; oD == oDr[0]) so that the chain
  // does not wait for a weight gradient before it overwrites its operand; gi_net_backward[_phase] joins before it returns.
  int64_t oDr[3] = {-1, -1, -1};
  // per slot: one 64-bit sign word per pixel of the first layer's activation (op_c1_gather), what the fused LeakyReLU backward in
  // the second layer's input-gradient GEMM reads instead of the activation itself (IgemmArgs::mask_bits); slot_bits1: written
  int64_t oBits1 = -1;
  std::vector<int> slot_bits1;
  int64_t oC1w = -1, c1w_floats = 0;   // shared: per-workgroup partial sums of the fused first-layer weight gradient (IgemmArgs::c1w_part)
  hipStream_t st2 = nullptr;
  hipEvent_t ev_dz = nullptr, ev_wg[3] = {nullptr, nullptr, nullptr};
  bool wg_busy[3] = {false, false, false};
  int dz_i = 0;
  bool side = false;                 // (transient) the backward in progress uses st2
  // train-mode BatchNorm statistics without a reduction launch: the GEMM epilogues add their tile sums into exact
  // per-channel accumulators (stat_acc.h), the normalisation pass derives scale / shift from them itself (and the backward
  // reductions likewise). Regions per BatchNorm layer: BN::acc_off; all zero after gi_net_bind, then kept zero by the
  // consuming passes. GI_BN_ACC=0 restores partial rows + finalize / sums launches.
  int64_t oAcc = -1, acc_words = 0;
  int use_acc = 1;
  std::vector<BwdFuse> bwd_pending;  // per BatchNorm id: a reduction planned / done by the producing GEMM, waiting for act_bn_bwd
  int64_t oTickets = -1;             // split-K fix-up counters (IgemmArgs::tickets): zero at bind, zero after every launch
  // gradient-penalty scratch (patchgan): stacked 2n tensors, see patchgan_gradient_penalty
  int64_t oA2[5] = {-1, -1, -1, -1, -1}, oG2[5] = {-1, -1, -1, -1, -1}, oTX[5] = {-1, -1, -1, -1, -1};
  int64_t oD2 = -1, oTZ = -1, oGimg = -1, oVimg = -1, oGPs = -1, oGPpart = -1, oGPsums = -1, oTh = -1;
  int gp_slot = -1;
  std::vector<int> slot_n, slot_train, slot_groups;
  std::vector<std::vector<const uint8_t*>> ext_mask;  // [slot][level]

  float* params = nullptr;
  float* grads = nullptr;
  float* buffers = nullptr;
  char* ws = nullptr;
  bool bound = false;

  size_t tsz() const { return gi_dtype_size(dtype); }
  char* slot(int s, int64_t off) const { return ws + slot_base + (int64_t)s * slot_bytes + off; }
  char* shared(int64_t off) const { return ws + off; }
};

namespace {

void add_tensor(gi_net* net, const std::string& name, int kind, std::initializer_list<int64_t> shape, int64_t* off_out) {
  TensorDesc t;
  t.name = name;
  t.kind = kind;
  t.ndim = (int)shape.size();
  t.numel = 1;
  int i = 0;
  for (int64_t s : shape) { t.shape[i++] = s; t.numel *= s; }
  for (; i < 4; ++i) t.shape[i] = 1;
  int64_t& ctr = (kind <= 1) ? net->n_params : net->n_buffers;
  t.offset = ctr;
  ctr += gi_align_up(t.numel, 64);
  *off_out = t.offset;
  net->tensors.push_back(t);
}

void add_bn(gi_net* net, const std::string& prefix, BN& bn, int c, std::vector<std::pair<std::string, BN*>>& later) {
  bn.c = c;
  add_tensor(net, prefix + ".weight", 1, {c}, &bn.gamma_off);
  add_tensor(net, prefix + ".bias", 1, {c}, &bn.beta_off);
  later.push_back({prefix, &bn});
}

int64_t max64(int64_t a, int64_t b) { return a > b ? a : b; }

// upper bound on partial-statistics rows any kernel may write for a (pixels, c) tensor
int64_t part_rows(int64_t pixels) { return max64(pixels / 64 + 8, 1100); }

}  // namespace

// =================================================================================================
// creation
// =================================================================================================
extern "C" int gi_unet_create_padded(gi_ctx* ctx, int num_downs, int ngf, int ch1, int out_c, int norm_kind, float dropout_p, int H, int W,
                                     int max_n, int dtype, int n_slots, gi_net** out) {
  GI_REQUIRE(out, "unet_create: null argument");  // ctx may be null: inventory-only handle
  GI_REQUIRE(norm_kind >= 0 && norm_kind <= 2, "unet_create: norm_kind=%d (0 batch, 1 instance, 2 none)", norm_kind);
  GI_REQUIRE(ch1 == 0 || (ch1 >= ngf && ch1 % 64 == 0), "unet_create: ch1=%d must be 0 or a multiple of 64 >= ngf", ch1);
  GI_REQUIRE(dtype == GI_F16 || dtype == GI_F32, "unet_create: dtype=%d", dtype);
  GI_REQUIRE(num_downs >= 5 && num_downs <= 9, "unet_create: num_downs=%d (supported 5..9)", num_downs);
  // the GEMM kernels take channel counts that are multiples of 64: every level must qualify. ngf = 32 does from level 2 on
  // (64, 128, 256, ...); its level 1 is then computed ch1 = 64 channels wide with the upper half zero (gi_unet_create_padded)
  if (ctx) GI_REQUIRE(ngf >= 32 && (2 * ngf) % 64 == 0 && ((ch1 ? ch1 : ngf) % 64 == 0),
                      "unet_create: ngf=%d ch1=%d: every level needs a multiple of 64 channels (ngf a multiple of 64, or of 32 with ch1)", ngf, ch1);
  else GI_REQUIRE(ngf % 8 == 0 && ngf >= 8, "unet_create: ngf=%d must be a multiple of 8 (inventory-only handle)", ngf);
  GI_REQUIRE(out_c >= 1 && out_c <= 64, "unet_create: out_c=%d (1..64)", out_c);
  // (powers of two take the halo-resident kernels; other multiples of 2^num_downs - 192, 384 - fall back to the gather kernels)
  GI_REQUIRE(H >= (1 << num_downs) && W >= (1 << num_downs) && H % (1 << num_downs) == 0 && W % (1 << num_downs) == 0,
             "unet_create: H=%d W=%d must be multiples of 2^num_downs = %d", H, W, 1 << num_downs);
  GI_REQUIRE(max_n >= 1 && n_slots >= 1 && n_slots <= 8, "unet_create: max_n=%d n_slots=%d", max_n, n_slots);
  gi_net* net = new gi_net();
  net->ctx = ctx; net->kind = 0; net->dtype = dtype; net->H = H; net->W = W; net->max_n = max_n; net->n_slots = n_slots;
  net->nd = num_downs; net->ngf = ngf; net->dropout_p = dropout_p; net->out_c = out_c; net->norm_kind = norm_kind;
  net->loss_scale = dtype == GI_F16 ? 65536.f : 1.f;
  const int nd = num_downs;
  net->ch.assign(nd + 1, 0); net->Hk.assign(nd + 1, 0); net->Wk.assign(nd + 1, 0);
  net->conv.assign(nd + 1, Conv()); net->up.assign(nd + 1, Conv());
  net->dnorm.assign(nd + 1, BN()); net->unorm.assign(nd + 1, BN());
  for (int k = 1; k <= nd; ++k) {
    int m = 1 << (k - 1);
    if (m > 8) m = 8;
    net->ch[k] = (k == 1 && ch1) ? ch1 : ngf * m;
    net->Hk[k] = H >> k;
    net->Wk[k] = W >> k;
  }
  // parameters in the reference's named_parameters() order (recursive nesting, networks.py:296-318)
  std::vector<std::pair<std::string, BN*>> bns;
  std::vector<std::string> prefix(nd + 2);
  prefix[1] = "model.model";
  struct Emit {
    gi_net* net; int nd; std::vector<std::string>& prefix; std::vector<std::pair<std::string, BN*>>& bns;
    void run(int k) {
      const std::string& p = prefix[k];
      const bool outer = (k == 1), inner = (k == nd);
      const std::string down = p + (outer ? ".0" : ".1");
      const std::string upn = p + ((outer || inner) ? ".3" : ".5");
      Conv& cv = net->conv[k];
      cv.ca = net->ch[k]; cv.cb = outer ? 1 : net->ch[k - 1];
      const bool inorm = net->norm_kind == 1;
      add_tensor(net, down + ".weight", 0, {cv.ca, cv.cb, 4, 4}, &cv.w_off);
      if (inorm) add_tensor(net, down + ".bias", 1, {cv.ca}, &cv.bias_off);
      if (!outer && !inner) {
        if (net->norm_kind == 0) add_bn(net, p + ".2", net->dnorm[k], net->ch[k], bns);
        else net->dnorm[k].c = net->ch[k];     // parameter-free (or absent) norm: only the channel count
      }
      if (!inner) {
        prefix[k + 1] = p + (outer ? ".1" : ".3") + ".model";
        run(k + 1);
      }
      Conv& uv = net->up[k];
      uv.ca = inner ? net->ch[k] : 2 * net->ch[k];
      uv.cb = outer ? net->out_c : net->ch[k - 1];
      add_tensor(net, upn + ".weight", 0, {uv.ca, uv.cb, 4, 4}, &uv.w_off);
      if (outer || inorm) add_tensor(net, upn + ".bias", 1, {uv.cb}, &uv.bias_off);
      if (!outer) {
        if (net->norm_kind == 0) add_bn(net, p + (inner ? ".4" : ".6"), net->unorm[k], net->ch[k - 1], bns);
        else net->unorm[k].c = net->ch[k - 1];
      }
    }
  } emit{net, nd, prefix, bns};
  emit.run(1);
  for (auto& pb : bns) {
    add_tensor(net, pb.first + ".running_mean", 2, {pb.second->c}, &pb.second->rmean_off);
    add_tensor(net, pb.first + ".running_var", 3, {pb.second->c}, &pb.second->rvar_off);
  }

  // ---- workspace layout
  const int64_t T = (int64_t)net->tsz();
  Arena& A = net->arena;
  for (int k = 2; k <= nd; ++k) {
    for (Conv* c : {&net->conv[k], &net->up[k]}) {
      const int64_t cnt = (int64_t)c->ca * 16 * c->cb;
      c->packed_off = (dtype == GI_F32) ? -2 : A.take(cnt * T);  // fp32: the master itself is [a][16*b]
      c->phase_off = A.take(cnt * T);
    }
  }
  for (int k = 2; k <= nd; ++k) {   // inference copies: conv in its packed form, transposed conv in its phase form
    if (k < nd) { net->conv[k].inf_off = A.take((int64_t)net->conv[k].ca * 16 * net->conv[k].cb * T); net->dnorm[k].inf_off = A.take(4 * net->dnorm[k].c * 4); }
    net->up[k].inf_off = A.take((int64_t)net->up[k].ca * 16 * net->up[k].cb * T);
    net->unorm[k].inf_off = A.take(4 * net->unorm[k].c * 4);
  }
  const int64_t N = max_n;
  net->ogC.assign(nd + 1, -1); net->ogA.assign(nd + 1, -1);
  int64_t maxD = 0, maxPart = 0, maxSplit = 0, maxc = 0;
  for (int k = 1; k <= nd - 1; ++k) {
    const int64_t pix = N * net->Hk[k] * net->Wk[k];
    net->ogC[k] = A.take(pix * 2 * net->ch[k] * T);
    net->ogA[k] = A.take(pix * net->ch[k] * T);
    maxD = max64(maxD, pix * net->ch[k] * T);
    maxPart = max64(maxPart, part_rows(pix) * 2 * 2 * net->ch[k]);
    maxc = max64(maxc, 2 * net->ch[k]);
    if (pix <= 32768) maxSplit = max64(maxSplit, pix * 2 * net->ch[k] * 4);
  }
  {
    const int64_t pix = N * net->Hk[nd] * net->Wk[nd];
    net->ogE = A.take(pix * net->ch[nd] * T);
    maxD = max64(maxD, pix * net->ch[nd] * T);
    maxSplit = max64(maxSplit, pix * net->ch[nd] * 4);
  }
  net->oD = net->oDr[0] = A.take(maxD);
  net->oDr[1] = A.take(maxD);
  net->oDr[2] = A.take(maxD);
  if (nd >= 2) {   // workgroups of d2's input-gradient GEMM (256-pixel patches of level 2, two py phases) x 64 x 16 sums
    net->c1w_floats = ((N * net->Hk[2] * net->Wk[2] / 256 + 8) * 2 + 64) * 1024;   // + 64 rows: op_c1_wgrad_reduce's first stage
    net->oC1w = A.take(net->c1w_floats * 4);
  }
  net->oG0 = A.take(N * H * W * 4 * out_c);
  if (out_c > 1) {   // u1 on the generic kernels: weights [a][16][b] with b zero-padded to 64, NHWC staging buffer
    const int64_t cnt = (int64_t)2 * net->ch[1] * 16 * 64;
    net->oUp1Pad = A.take(cnt * 4);
    net->oUp1Packed = (dtype == GI_F32) ? -2 : A.take(cnt * T);
    net->oUp1Phase = A.take(cnt * T);
    net->oU1 = A.take(N * H * W * 64 * T);
  }
  net->oCol = A.take(N * (H / 2) * (W / 2) * 16 * 2 * (out_c == 4 ? 4 : 1));   // (x4: the 4-channel head's col buffer)
  net->part_floats = maxPart;
  net->oPart = A.take(maxPart * 4);
  net->oSums = A.take(8 * maxc * 4);   // [8][c]: sums + apply coefficients
  // split-K scratch: one fp32 buffer per split (split x output ~ 384 tiles of 128 x 128) for the deterministic path
  if (maxSplit > 0) maxSplit = max64(maxSplit, (int64_t)400 * 128 * 128 * 4);
  net->split_bytes = maxSplit;
  net->oSplit = A.take(maxSplit);
  for (int k = 2; k <= nd; ++k) {
    net->wg_bytes = max64(net->wg_bytes, op_wgrad_scratch_bytes(dtype, max_n, net->Hk[k], net->Wk[k], net->conv[k].ca, net->conv[k].cb));
    net->wg_bytes = max64(net->wg_bytes, op_wgrad_scratch_bytes(dtype, max_n, net->Hk[k], net->Wk[k], net->up[k].ca, net->up[k].cb));
  }
  if (net->wg_bytes > 0) net->oWg = A.take(net->wg_bytes);
  // per-slot
  Arena S;
  net->oC.assign(nd + 1, -1); net->oR.assign(nd + 1, -1); net->oU.assign(nd + 1, -1); net->oMask.assign(nd + 1, -1);
  for (int k = 1; k <= nd - 1; ++k) net->oC[k] = S.take(N * net->Hk[k] * net->Wk[k] * 2 * net->ch[k] * T);
  for (int k = 2; k <= nd - 1; ++k) net->oR[k] = S.take(N * net->Hk[k] * net->Wk[k] * net->ch[k] * T);
  net->oE = S.take(N * net->Hk[nd] * net->Wk[nd] * net->ch[nd] * T);
  for (int k = 2; k <= nd; ++k) net->oU[k] = S.take(N * net->Hk[k - 1] * net->Wk[k - 1] * net->ch[k - 1] * T);
  for (int k = 5; k <= nd - 1; ++k) net->oMask[k] = S.take(N * net->Hk[k - 1] * net->Wk[k - 1] * net->ch[k - 1]);
  net->oOut = S.take(N * H * W * 4 * out_c);
  net->oX = S.take(N * H * W * 4);
  net->oBits1 = S.take(N * net->Hk[1] * net->Wk[1] * 8 + 16);
  int64_t stat_floats = 0;
  int nbn = 0;
  for (int k = 1; k <= nd; ++k)
    for (BN* b : {&net->dnorm[k], &net->unorm[k]})
      if (b->c) { b->stat_off = stat_floats; stat_floats += 4 * b->c; b->id = nbn++; }
  net->oStats = S.take(stat_floats * 4);
  if (norm_kind == 1) {
    int64_t in_floats = 0;
    for (int k = 1; k <= nd; ++k)
      for (BN* b : {&net->dnorm[k], &net->unorm[k]})
        if (b->c) { b->in_off = in_floats; in_floats += 2 * N * b->c; }
    net->oInStats = S.take(in_floats * 4);
  }
  net->slot_bytes = S.size;
  net->slot_base = A.take(S.size * n_slots);
  for (int k = 1; k <= nd; ++k)
    for (BN* b : {&net->dnorm[k], &net->unorm[k]})
      if (b->c && norm_kind == 0) { b->acc_off = net->acc_words; net->acc_words += 4 * b->acc_block(); }
  net->oAcc = A.take(net->acc_words * 8);
  net->oTickets = A.take((int64_t)GI_IGEMM_TICKETS * 4);
  net->use_acc = gi_opt(GI_OPT_BN_ACC);   // GI_BN_ACC=0: partial rows + finalize / sums launches
  net->slot_n.assign(n_slots, 0);
  net->slot_train.assign(n_slots, 0);
  net->slot_fused_u2.assign(n_slots, 0);
  net->slot_bits1.assign(n_slots, 0);
  net->slot_inference.assign(n_slots, 0);
  net->eval_gen.assign(n_slots, std::vector<uint64_t>(nbn, 0));
  net->bwd_pending.assign(nbn, BwdFuse());
  net->fuse_head = gi_opt(GI_OPT_FUSE_HEAD);   // GI_FUSE_HEAD=0: materialise the last decoder level
  net->ext_mask.assign(n_slots, std::vector<const uint8_t*>(nd + 1, nullptr));
  *out = net;
  return GI_OK;
}

extern "C" int gi_unet_create_norm(gi_ctx* ctx, int num_downs, int ngf, int out_c, int norm_kind, float dropout_p, int H, int W,
                                   int max_n, int dtype, int n_slots, gi_net** out) {
  return gi_unet_create_padded(ctx, num_downs, ngf, 0, out_c, norm_kind, dropout_p, H, W, max_n, dtype, n_slots, out);
}

extern "C" int gi_unet_create_ex(gi_ctx* ctx, int num_downs, int ngf, int out_c, float dropout_p, int H, int W, int max_n,
                                 int dtype, int n_slots, gi_net** out) {
  return gi_unet_create_norm(ctx, num_downs, ngf, out_c, 0, dropout_p, H, W, max_n, dtype, n_slots, out);
}

extern "C" int gi_unet_create(gi_ctx* ctx, int num_downs, int ngf, float dropout_p, int H, int W, int max_n, int dtype,
                              int n_slots, gi_net** out) {
  return gi_unet_create_ex(ctx, num_downs, ngf, 1, dropout_p, H, W, max_n, dtype, n_slots, out);
}

extern "C" int gi_patchgan_create(gi_ctx* ctx, int H, int W, int sigmoid, int max_n, int dtype, int n_slots, gi_net** out) {
  GI_REQUIRE(out, "patchgan_create: null argument");  // ctx may be null: inventory-only handle
  GI_REQUIRE(dtype == GI_F16 || dtype == GI_F32, "patchgan_create: dtype=%d", dtype);
  GI_REQUIRE(H >= 64 && W >= 64 && H % 16 == 0 && W % 16 == 0, "patchgan_create: H=%d W=%d must be multiples of 16, >= 64", H, W);
  GI_REQUIRE(max_n >= 1 && n_slots >= 1 && n_slots <= 8, "patchgan_create: max_n=%d n_slots=%d", max_n, n_slots);
  gi_net* net = new gi_net();
  net->ctx = ctx; net->kind = 1; net->dtype = dtype; net->H = H; net->W = W; net->max_n = max_n; net->n_slots = n_slots;
  net->sigmoid = sigmoid;
  net->loss_scale = dtype == GI_F16 ? 65536.f : 1.f;
  net->Hh = H / 16; net->Wh = W / 16;
  net->P = (net->Hh - 3) * (net->Wh - 3);
  const int chans[5] = {1, 64, 128, 256, 512};
  const int conv_idx[5] = {0, 0, 2, 5, 8};
  const int bn_idx[5] = {0, 0, 3, 6, 9};
  std::vector<std::pair<std::string, BN*>> bns;
  for (int i = 1; i <= 4; ++i) {
    Conv& c = net->dconv[i];
    c.ca = chans[i]; c.cb = chans[i - 1];
    add_tensor(net, "model." + std::to_string(conv_idx[i]) + ".weight", 0, {c.ca, c.cb, 4, 4}, &c.w_off);
    if (i >= 2) add_bn(net, "model." + std::to_string(bn_idx[i]), net->dbn[i], chans[i], bns);
  }
  add_tensor(net, "model.11.weight", 0, {1, 512, 4, 4}, &net->w5_off);
  add_tensor(net, "model.13.weight", 1, {1, net->P}, &net->wl_off);
  add_tensor(net, "model.13.bias", 1, {1}, &net->bl_off);
  for (auto& pb : bns) {
    add_tensor(net, pb.first + ".running_mean", 2, {pb.second->c}, &pb.second->rmean_off);
    add_tensor(net, pb.first + ".running_var", 3, {pb.second->c}, &pb.second->rvar_off);
  }
  const int64_t T = (int64_t)net->tsz();
  Arena& A = net->arena;
  for (int i = 2; i <= 4; ++i) {
    Conv& c = net->dconv[i];
    const int64_t cnt = (int64_t)c.ca * 16 * c.cb;
    c.packed_off = (dtype == GI_F32) ? -2 : A.take(cnt * T);
    c.phase_off = A.take(cnt * T);
  }
  const int64_t N = max_n;
  int64_t maxD = 0, maxPart = 0, maxSplit = 0;
  net->ogA.assign(5, -1);
  for (int i = 1; i <= 4; ++i) {
    const int64_t pix = N * (H >> i) * (W >> i);
    net->ogA[i] = A.take(pix * chans[i] * T);
    maxD = max64(maxD, pix * chans[i] * T);
    maxPart = max64(maxPart, part_rows(pix) * 2 * chans[i] * 3);   // x3: room for the per-group column pass (BN groups)
    if (pix <= 32768) maxSplit = max64(maxSplit, pix * chans[i] * 4);
  }
  net->oD = net->oDr[0] = A.take(maxD);
  net->oDr[1] = A.take(maxD);
  net->oDr[2] = A.take(maxD);
  net->c1w_floats = ((N * (H >> 2) * (W >> 2) / 256 + 8) * 2 + 64) * 1024;   // + 64 rows: op_c1_wgrad_reduce's first stage
  net->oC1w = A.take(net->c1w_floats * 4);
  net->part_floats = maxPart;
  net->oPart = A.take(maxPart * 4);
  net->oSums = A.take(2 * 8 * 512 * 4);   // [groups][8][c]: sums + apply coefficients
  if (maxSplit > 0) maxSplit = max64(maxSplit, (int64_t)400 * 128 * 128 * 4);
  net->split_bytes = maxSplit;
  net->oSplit = A.take(maxSplit > 0 ? maxSplit : 16);
  for (int i = 2; i <= 4; ++i)
    net->wg_bytes = max64(net->wg_bytes, op_wgrad_scratch_bytes(dtype, max_n, H >> i, W >> i, net->dconv[i].ca, net->dconv[i].cb));
  if (net->wg_bytes > 0) net->oWg = A.take(net->wg_bytes);
  net->oDh = A.take(N * net->P * 4);
  net->hw_bytes = op_head_scratch_bytes((int)N, net->Hh, net->Wh);
  net->oHw = A.take(net->hw_bytes);
  if (net->Hh * net->Wh > 256) { net->ht_bytes = N * net->Hh * net->Wh * 16 * 4; net->oHt = A.take(net->ht_bytes); }
  net->oCol = A.take(N * (H / 2) * (W / 2) * 16 * 2);
  {   // gradient penalty (fp32 critics): tangent / stacked-gradient tensors
    int64_t maxl = 0;
    for (int i = 1; i <= 4; ++i) {
      const int64_t e = N * (H >> i) * (W >> i) * chans[i] * T;
      net->oA2[i] = A.take(2 * e);
      net->oG2[i] = A.take(2 * e);
      net->oTX[i] = A.take(e);
      maxl = max64(maxl, e);
    }
    net->oD2 = A.take(2 * maxl);
    net->oTZ = A.take(maxl);
    net->oGimg = A.take(N * H * W * 4);
    net->oVimg = A.take(N * H * W * 4);
    net->oGPs = A.take((N + 16) * 4 * 2);
    net->oGPpart = A.take((int64_t)1100 * 5 * 512 * 4);
    net->oGPsums = A.take(5 * 512 * 4);
    net->oTh = A.take(N * net->P * 4 * 2);
  }
  Arena S;
  for (int i = 1; i <= 4; ++i) net->oA[i] = S.take(N * (H >> i) * (W >> i) * chans[i] * T);
  for (int i = 2; i <= 4; ++i) net->oRd[i] = S.take(N * (H >> i) * (W >> i) * chans[i] * T);
  net->oHh = S.take(N * net->P * 4);
  net->oOut = S.take(N * 4);
  net->oX = S.take(N * H * W * 4);
  net->oBits1 = S.take(N * (H >> 1) * (W >> 1) * 8 + 16);
  int64_t stat_floats = 0;
  for (int i = 2; i <= 4; ++i) { net->dbn[i].stat_off = stat_floats; stat_floats += 2 * 4 * net->dbn[i].c; net->dbn[i].id = i - 2; }   // x2: BN groups
  net->eval_gen.assign(n_slots + 1, std::vector<uint64_t>(3, 0));
  net->bwd_pending.assign(3, BwdFuse());
  net->oStats = S.take(stat_floats * 4);
  net->slot_bytes = S.size;
  for (int i = 2; i <= 4; ++i) { net->dbn[i].acc_off = net->acc_words; net->acc_words += 4 * net->dbn[i].acc_block(); }
  net->oAcc = A.take(net->acc_words * 8);
  net->oTickets = A.take((int64_t)GI_IGEMM_TICKETS * 4);
  net->use_acc = gi_opt(GI_OPT_BN_ACC);   // GI_BN_ACC=0: partial rows + finalize / sums launches
  net->gp_slot = n_slots;                      // one private activation set for the gradient penalty
  net->slot_base = A.take(S.size * (n_slots + 1));
  net->slot_n.assign(n_slots + 1, 0);
  net->slot_train.assign(n_slots + 1, 0);
  net->slot_fused_u2.assign(n_slots + 1, 0);
  net->slot_bits1.assign(n_slots + 1, 0);
  net->fuse_head = gi_opt(GI_OPT_FUSE_HEAD);
  net->slot_groups.assign(n_slots + 1, 1);
  *out = net;
  return GI_OK;
}

extern "C" int gi_net_destroy(gi_net* net) {
  if (net) {
    if (net->st2) { (void)hipStreamSynchronize(net->st2); (void)hipStreamDestroy(net->st2); }
    if (net->ev_dz) (void)hipEventDestroy(net->ev_dz);
    for (hipEvent_t e : net->ev_wg) if (e) (void)hipEventDestroy(e);
  }
  delete net;
  return GI_OK;
}

// =================================================================================================
// inventory / binding
// =================================================================================================
extern "C" int gi_net_tensor_count(gi_net* net) { return net ? (int)net->tensors.size() : GI_ERR_INVALID; }

extern "C" int gi_net_tensor_desc(gi_net* net, int index, char* name, int name_cap, int* kind, int64_t* shape4, int* ndim,
                                  int64_t* offset, int64_t* numel) {
  GI_REQUIRE(net && index >= 0 && index < (int)net->tensors.size(), "tensor_desc: index %d out of range", index);
  const TensorDesc& t = net->tensors[index];
  if (name && name_cap > 0) {
    strncpy(name, t.name.c_str(), name_cap - 1);
    name[name_cap - 1] = 0;
  }
  if (kind) *kind = t.kind;
  if (shape4) for (int i = 0; i < 4; ++i) shape4[i] = t.shape[i];
  if (ndim) *ndim = t.ndim;
  if (offset) *offset = t.offset;
  if (numel) *numel = t.numel;
  return GI_OK;
}
extern "C" int64_t gi_net_param_floats(gi_net* net) { return net ? net->n_params : GI_ERR_INVALID; }
extern "C" int64_t gi_net_buffer_floats(gi_net* net) { return net ? net->n_buffers : GI_ERR_INVALID; }
extern "C" int64_t gi_net_workspace_bytes(gi_net* net) { return net ? net->arena.size : GI_ERR_INVALID; }

extern "C" int gi_net_bind(gi_net* net, float* params, float* grads, float* buffers, void* workspace, int64_t workspace_bytes) {
  GI_REQUIRE(net && params && grads && workspace, "net_bind: null pointer");
  GI_REQUIRE(buffers || net->n_buffers == 0, "net_bind: null buffers for a net with %lld buffer floats", (long long)net->n_buffers);
  GI_REQUIRE(net->ctx != nullptr, "net_bind: handle was created without a context (inventory only)");
  GI_REQUIRE(workspace_bytes >= net->arena.size, "net_bind: workspace %lld < required %lld", (long long)workspace_bytes,
             (long long)net->arena.size);
  GI_REQUIRE(((uintptr_t)workspace & 255) == 0 && ((uintptr_t)params & 15) == 0, "net_bind: workspace must be 256-byte aligned");
  net->params = params; net->grads = grads; net->buffers = buffers; net->ws = (char*)workspace;
  net->bound = true;
  if (net->acc_words > 0) GI_HIP(hipMemsetAsync(net->shared(net->oAcc), 0, (size_t)net->acc_words * 8, net->ctx->stream));
  if (net->oTickets >= 0) GI_HIP(hipMemsetAsync(net->shared(net->oTickets), 0, (size_t)GI_IGEMM_TICKETS * 4, net->ctx->stream));
  return GI_OK;
}

namespace {
const void* packed_ptr(const gi_net* net, const Conv& c) {
  return c.packed_off == -2 ? (const void*)(net->params + c.w_off) : (const void*)net->shared(c.packed_off);
}
const void* phase_ptr(const gi_net* net, const Conv& c) { return net->shared(c.phase_off); }

// [rows][b] -> [rows][bp] zero-padded (the outermost up-convolution of a multi-channel generator)
__global__ void __launch_bounds__(256) pad_b_kernel(const float* __restrict__ src, float* __restrict__ dst, int64_t rows, int b, int bp) {
  const int64_t total = rows * bp;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int j = (int)(i % bp);
    dst[i] = j < b ? src[(i / bp) * b + j] : 0.f;
  }
}

}  // namespace

extern "C" int gi_net_sync_weights(gi_net* net) {
  GI_REQUIRE(net && net->bound, "sync_weights: net not bound");
  ++net->affine_gen;   // parameters (and possibly running statistics) were written from outside
  PackJobs P;
  P.n = 0;
  auto add = [&](const Conv& c) {
    PackJob& J = P.j[P.n++];
    J.w = net->params + c.w_off; J.ca = c.ca; J.cb = c.cb; J.tile0 = 0; J.scale_on_b = 0; J.scale = nullptr;
    J.packed = c.packed_off >= 0 ? (void*)net->shared(c.packed_off) : nullptr;
    J.phase = net->shared(c.phase_off);
  };
  if (net->kind == 0) {
    for (int k = 2; k <= net->nd; ++k) {
      if (P.n + 2 > 16) { GI_TRY(op_pack_weights_batch(net->ctx->stream, net->dtype, P)); P.n = 0; }
      add(net->conv[k]);
      add(net->up[k]);
    }
    GI_TRY(op_pack_weights_batch(net->ctx->stream, net->dtype, P));
    if (net->out_c > 1) {
      const int ca = net->up[1].ca;
      float* pad = (float*)net->shared(net->oUp1Pad);
      hipLaunchKernelGGL(pad_b_kernel, dim3(256), dim3(256), 0, net->ctx->stream, net->params + net->up[1].w_off, pad, (int64_t)ca * 16,
                         net->out_c, 64);
      GI_LAUNCH_CHECK();
      GI_TRY(op_pack_weights(net->ctx->stream, net->dtype, pad, ca, 64, net->oUp1Packed >= 0 ? (void*)net->shared(net->oUp1Packed) : nullptr,
                             net->shared(net->oUp1Phase)));
    }
  } else {
    for (int i = 2; i <= 4; ++i) add(net->dconv[i]);
    GI_TRY(op_pack_weights_batch(net->ctx->stream, net->dtype, P));
  }
  return GI_OK;
}
extern "C" int gi_net_set_train(gi_net* net, int train) {
  GI_REQUIRE(net, "set_train: null");
  if (net->train != (train ? 1 : 0)) ++net->affine_gen;
  net->train = train ? 1 : 0;
  return GI_OK;
}
extern "C" int gi_net_set_inference(gi_net* net, int inference) {
  GI_REQUIRE(net, "set_inference: null");
  net->inference = inference ? 1 : 0;
  return GI_OK;
}
extern "C" int gi_net_set_bn_groups(gi_net* net, int groups) {
  GI_REQUIRE(net && net->kind == 1, "set_bn_groups: discriminator handle required");
  GI_REQUIRE(groups == 1 || groups == 2, "set_bn_groups: groups=%d (1 or 2)", groups);
  net->bn_groups = groups;
  return GI_OK;
}
extern "C" int gi_net_set_loss_scale(gi_net* net, float scale) {
  GI_REQUIRE(net && scale > 0.f, "set_loss_scale: scale must be > 0");
  net->loss_scale = scale;
  return GI_OK;
}
extern "C" int gi_net_set_dropout_seed(gi_net* net, uint64_t seed) {
  GI_REQUIRE(net, "set_dropout_seed: null");
  net->drop_seed = seed;
  net->drop_counter = 0;
  return GI_OK;
}
extern "C" int gi_net_dropout_mask(gi_net* net, int slot, int level, uint8_t* out_nchw, int64_t count) {
  GI_REQUIRE(net && net->kind == 0 && net->bound, "dropout_mask: generator handle required");
  GI_REQUIRE(slot >= 0 && slot < net->n_slots && level >= 5 && level <= net->nd - 1, "dropout_mask: slot=%d level=%d", slot, level);
  const int n = net->slot_n[slot], c = net->ch[level - 1], hw = net->Hk[level - 1] * net->Wk[level - 1];
  GI_REQUIRE(count == (int64_t)n * c * hw, "dropout_mask: count %lld != %lld", (long long)count, (long long)n * c * hw);
  return op_mask_nchw_to_nhwc(net->ctx->stream, (const uint8_t*)net->slot(slot, net->oMask[level]), out_nchw, n, c, hw, 0);
}
extern "C" int gi_net_set_dropout_mask(gi_net* net, int slot, int level, const uint8_t* mask_nchw) {
  GI_REQUIRE(net && net->kind == 0, "set_dropout_mask: generator handle required");
  GI_REQUIRE(slot >= 0 && slot < net->n_slots && level >= 5 && level <= net->nd - 1, "set_dropout_mask: slot=%d level=%d", slot, level);
  net->ext_mask[slot][level] = mask_nchw;
  return GI_OK;
}

// =================================================================================================
// forward / backward
// =================================================================================================
namespace {

struct BNPtrs { float *scale, *shift, *mean, *inv; };
BNPtrs bn_ptrs(const gi_net* net, int slot, const BN& b, int group = 0) {
  float* base = (float*)net->slot(slot, net->oStats) + b.stat_off + (int64_t)group * 4 * b.c;
  return {base, base + b.c, base + 2 * b.c, base + 3 * b.c};
}

// raw (pixels,c) conv output with per-tile partial statistics -> normalise + activation into dst.
// With bn_groups = g the batch is g consecutive groups of pixels/g pixels, each normalised with its OWN batch
// statistics (the reference calls the critic separately on the real and on the fake batch: two BatchNorm
// populations, running statistics updated group by group in that order).
// apply = false: statistics / scale / shift only (a consumer applies the affine map itself, C1Affine)
// Statistics plan of one BatchNorm layer, fixed BEFORE its GEMM runs: with `use` the GEMM adds its tile sums to the
// layer's exact accumulators (IgemmArgs::stat_acc) and bn_forward needs no reduction launch.
struct StatPlan { unsigned long long* acc = nullptr; unsigned long long* zero_next = nullptr; int zero_words = 0; int pg = 0; int reps = 1; bool use = false; };
// (`use` is cleared by the igemm wrapper when the kernel that ran put its statistics into partial rows instead)
// replicas for `adders` tiles adding to one block: about 256 requests per 64-byte line (~3 us of queueing at the atomic unit)
int stat_reps_for(int64_t adders) {
  int r = 1;
  while (r < GI_STAT_MAXREP && adders > 256 * r) r <<= 1;
  return r;
}
// tile_rows: an upper bound on the partial rows the GEMM's tiles would write (tiles along M x sub-pixel phases)
StatPlan stat_plan(const gi_net* net, const BN& b, int64_t gemm_rows, int64_t tile_rows, int train) {
  StatPlan sp;
  const int g = net->kind == 1 ? net->bn_groups : 1;
  // two populations: every tile (256 / 128 rows, or a patch of one image) must lie inside one of them
  if (!train || !net->use_acc || b.acc_off < 0 || (g == 2 && (gemm_rows % 2 != 0 || (gemm_rows / 2) % 256 != 0))) return sp;
  unsigned long long* base = (unsigned long long*)net->shared(net->oAcc) + b.acc_off;
  sp.use = true;
  sp.reps = stat_reps_for(tile_rows);
  sp.acc = base + b.fwd_par * b.acc_block();
  sp.zero_next = base + (b.fwd_par ^ 1) * b.acc_block();
  sp.zero_words = (int)gi_stat_block_words(b.c, b.dirty[b.fwd_par ^ 1]);   // (the state moves when the plan is consumed: bn_forward)
  sp.pg = g == 2 ? (int)(gemm_rows / 2) : 0;
  return sp;
}

// the accumulator-path arguments of a layer's normalisation (pg pixels per population, g populations), and the state change of
// the layer's ping-pong accumulator regions once a launch that consumes them has been issued
BnAccArgs bn_acc_args(const gi_net* net, int slot, const BN& b, const StatPlan& sp, int64_t pg, int g) {
  BNPtrs p = bn_ptrs(net, slot, b, 0);
  BnAccArgs a;
  a.acc = sp.acc; a.gamma = net->params + b.gamma_off; a.beta = net->params + b.beta_off;
  a.running_mean = net->buffers + b.rmean_off; a.running_var = net->buffers + b.rvar_off;
  a.scale = p.scale; a.shift = p.shift; a.save_mean = p.mean; a.save_invstd = p.inv;
  a.count = pg; a.momentum = 0.1f; a.eps = 1e-5f; a.groups = g; a.out_stride = 4 * b.c;
  a.zero_next = sp.zero_words > 0 ? sp.zero_next : nullptr; a.zero_words = sp.zero_words;
  a.reps = sp.reps;
  return a;
}
void bn_acc_commit(gi_net* net, int slot, const BN& b, const StatPlan& sp) {
  b.dirty[b.fwd_par ^ 1] = 0;
  b.dirty[b.fwd_par] = sp.reps;
  b.fwd_par ^= 1;
  if (b.id >= 0) net->eval_gen[slot][b.id] = 0;
}

// drop_p > 0 with the accumulator path: the keep-mask is drawn inside the normalisation pass (seed drop_seed) and
// stored at `drop`; otherwise `drop` (if any) is read.
int bn_forward(gi_net* net, int slot, const BN& b, const void* raw, int64_t pixels, int ntiles, void* dst, int ldy,
               int coffy, int act, const uint8_t* drop, float drop_scale, int train, bool apply = true,
               const StatPlan* sp = nullptr, float drop_p = 0.f, uint64_t drop_seed = 0,
               const void* side_src = nullptr, void* side_dst = nullptr, int64_t side_bytes = 0) {   // side copy: accumulator path only
  hipStream_t st = net->ctx->stream;
  const int g = net->kind == 1 ? net->bn_groups : 1;
  const int64_t pg = pixels / g;
  const size_t T = net->tsz();
  if (sp && sp->use) {
    const BnAccArgs a = bn_acc_args(net, slot, b, *sp, pg, g);
    bn_acc_commit(net, slot, b, *sp);
    if (apply) return op_bn_apply_acc(st, net->dtype, raw, dst, pixels, b.c, ldy, coffy, act, (uint8_t*)drop, drop_scale, drop_seed, drop_p, a,
                                      side_src, side_dst, side_bytes);
    return op_bn_finalize_acc(st, b.c, a);
  }
  // the GEMM epilogue's partial rows can be split between the groups when a group is a whole number of tiles
  // (tile heights are 256, 128 or a power of two <= 64); otherwise each group's statistics come from a column pass
  const bool aligned = g == 1 || (pg % 256 == 0 && ntiles % g == 0);
  if (g == 2 && aligned) {   // both populations in one finalize and one apply launch
    BNPtrs p = bn_ptrs(net, slot, b, 0);
    const int rows = ntiles / 2;
    GI_TRY(op_bn_finalize(st, (const float*)net->shared(net->oPart), rows, b.c, pg, net->params + b.gamma_off, net->params + b.beta_off,
                          net->buffers + b.rmean_off, net->buffers + b.rvar_off, p.scale, p.shift, p.mean, p.inv, train, 0.1f, 1e-5f, 2,
                          (int64_t)rows * 2 * b.c, 4 * b.c));
    if (apply) GI_TRY(op_bn_apply(st, net->dtype, raw, dst, pixels, b.c, ldy, coffy, p.scale, p.shift, act, drop, drop_scale, pg, 4 * b.c));
    return GI_OK;
  }
  for (int j = 0; j < g; ++j) {
    BNPtrs p = bn_ptrs(net, slot, b, j);
    const char* rj = (const char*)raw + (int64_t)j * pg * b.c * T;
    const float* part = (const float*)net->shared(net->oPart);
    int rows = ntiles / g;
    if (aligned) {
      part += (int64_t)j * rows * 2 * b.c;
    } else {
      float* scratch = (float*)net->shared(net->oPart) + (int64_t)ntiles * 2 * b.c;
      GI_REQUIRE((int64_t)ntiles * 2 * b.c + (pg / 32 + 8) * 2 * b.c <= net->part_floats, "internal: partials buffer too small for a column pass");
      GI_TRY(op_col_stats(st, net->dtype, rj, pg, b.c, scratch, &rows));
      part = scratch;
    }
    const bool cached = !train && g == 1 && b.id >= 0 && net->eval_gen[slot][b.id] == net->affine_gen;
    if (!cached) {
      GI_TRY(op_bn_finalize(st, part, rows, b.c, pg, net->params + b.gamma_off, net->params + b.beta_off, net->buffers + b.rmean_off,
                            net->buffers + b.rvar_off, p.scale, p.shift, p.mean, p.inv, train, 0.1f, 1e-5f));
      if (b.id >= 0) net->eval_gen[slot][b.id] = (!train && g == 1) ? net->affine_gen : 0;
    }
    if (apply)
      GI_TRY(op_bn_apply(st, net->dtype, rj, (char*)dst + (int64_t)j * pg * ldy * T, pg, b.c, ldy, coffy, p.scale, p.shift, act,
                         drop ? drop + (int64_t)j * pg * b.c : nullptr, drop_scale));
  }
  return GI_OK;
}

// pixels: rows of the BatchNorm'd tensor; gemm_tiles: workgroups of the producing GEMM that will add (tiles x phases)
BwdFuse bwd_fuse_plan(gi_net* net, int slot, const BN& bn, const void* x, int64_t pixels, int64_t gemm_tiles, float slope) {
  BwdFuse f;
  const int on = gi_opt(GI_OPT_BN_BWD_FUSE);   // GI_BN_BWD_FUSE=0: reduce pass as a separate launch
  const int g = net->kind == 1 ? net->bn_groups : 1;
  if (!on || !net->use_acc || bn.acc_off < 0 || net->bwd_eval || net->dtype != GI_F16) return f;
  const int64_t pg = pixels / g;
  if (g == 2) {   // the same condition under which act_bn_bwd reduces both populations in one launch
    const int rpb = op_bwd_rows_per_block(pg);
    if (pixels % 2 != 0 || pg % rpb != 0 || (2 * (pg / rpb)) * 2 * bn.c > net->part_floats) return f;
  }
  const int64_t blocks = (pg + op_bwd_rows_per_block(pg) - 1) / op_bwd_rows_per_block(pg);
  unsigned long long* base = (unsigned long long*)net->shared(net->oAcc) + bn.acc_off + 2 * bn.acc_block();
  f.planned = true;
  f.groups = g;
  f.acc = base + bn.bwd_par * bn.acc_block();
  f.reps_gemm = stat_reps_for(gemm_tiles);
  f.reps_reduce = stat_reps_for(blocks);
  const int other = 2 + (bn.bwd_par ^ 1);
  f.zero_words = (int)gi_stat_block_words(bn.c, bn.dirty[other]);
  f.zero_next = f.zero_words > 0 ? base + (bn.bwd_par ^ 1) * bn.acc_block() : nullptr;
  bn.dirty[other] = 0;
  bn.dirty[2 + bn.bwd_par] = f.reps_gemm > f.reps_reduce ? f.reps_gemm : f.reps_reduce;
  bn.bwd_par ^= 1;
  BNPtrs p0 = bn_ptrs(net, slot, bn, 0);
  f.x = x; f.ldx = bn.c; f.scale = p0.scale; f.shift = p0.shift; f.mean = p0.mean; f.inv = p0.inv; f.stride = 4 * bn.c;
  f.slope = slope; f.pg = g == 2 ? pg : 0;
  return f;
}

// IgemmArgs::c1w_* for the igemm() wrapper: the first layer's weight gradient from the second layer's input-gradient GEMM
struct C1WFuse {
  const float* img = nullptr; float* part = nullptr; int64_t part_floats = 0; float scale = 1.f; int skip_out = 0;
  int applied = 0, blocks = 0;   // (returned)
};

// mask / ldmask / mask_slope / mask_applied: IgemmArgs::mask (activation backward fused into an input-gradient GEMM)
int igemm(gi_net* net, int phase, const void* in, int cin, int ldin, int coffin, const void* w, void* out, int cout,
          int ldout, int coffout, int n, int Hs, int Ws, int relu_in, int act_out, bool stats, int* ntiles, int relu_cend = 0,
          const void* mask = nullptr, int ldmask = 0, float mask_slope = 0.f, int* mask_applied = nullptr,
          const void* add = nullptr, int ldadd = 0, const float* bias = nullptr, StatPlan* sp = nullptr, BwdFuse* bf = nullptr,
          const IgemmFold* fold = nullptr, int* fold_applied = nullptr, const unsigned long long* mask_bits = nullptr, C1WFuse* c1w = nullptr) {
  IgemmArgs a;
  memset(&a, 0, sizeof(a));
  a.mask_bits = mask ? mask_bits : nullptr;
  if (c1w && a.mask_bits) { a.c1w_img = c1w->img; a.c1w_part = c1w->part; a.c1w_part_floats = c1w->part_floats; a.c1w_scale = c1w->scale; a.c1w_skip_out = c1w->skip_out; }
  if (stats && sp && sp->use) { a.stat_acc = sp->acc; a.stat_pg = sp->pg; a.stat_reps = sp->reps; a.fold = fold; }
  if (bf && bf->planned) {
    a.bwd_x = bf->x; a.bwd_ldx = bf->ldx; a.bwd_scale = bf->scale; a.bwd_shift = bf->shift; a.bwd_mean = bf->mean; a.bwd_inv = bf->inv;
    a.bwd_stride = bf->stride; a.bwd_slope = bf->slope; a.bwd_acc = bf->acc; a.bwd_reps = bf->reps_gemm; a.bwd_pg = bf->pg;
    a.bwd_c0 = bf->c0; a.bwd_c = bf->c;
  }
  a.relu_cend = relu_cend;
  a.mask = mask; a.ldmask = ldmask; a.coffmask = 0; a.mask_slope = mask_slope;
  a.add = add; a.ldadd = ldadd; a.coffadd = 0;
  a.in = in; a.w = w; a.out = out; a.bias = bias;
  a.partials = stats ? (float*)net->shared(net->oPart) : nullptr;
  a.ws = net->split_bytes > 0 ? (float*)net->shared(net->oSplit) : nullptr;
  a.ws_bytes = net->split_bytes;
  a.tickets = net->oTickets >= 0 ? (unsigned*)net->shared(net->oTickets) : nullptr;
  a.n = n; a.Hs = Hs; a.Ws = Ws;
  a.cin = cin; a.ldin = ldin; a.coffin = coffin;
  a.cout = cout; a.ldout = ldout; a.coffout = coffout;
  a.relu_in = relu_in; a.act_out = act_out;
  GI_TRY(op_igemm(net->ctx->stream, net->dtype, phase, a));
  if (mask_applied) *mask_applied = a.mask_applied;
  if (c1w) { c1w->applied = a.c1w_applied; c1w->blocks = a.c1w_blocks; }
  if (fold_applied) *fold_applied = a.fold_applied;
  if (bf && bf->planned) bf->applied = a.bwd_applied != 0;
  if (sp && sp->use && !a.stat_used) sp->use = false;
  if (ntiles) *ntiles = a.ntiles_out;
  if (stats) GI_REQUIRE((int64_t)a.ntiles_out * 2 * cout <= net->part_floats, "internal: partials buffer too small");
  return GI_OK;
}

int wgrad(gi_net* net, const void* S, int ca, int ldS, int coffS, int relu_S, const void* L, int cb, int ldL, int coffL,
          int n, int Hs, int Ws, float* dW) {
  WgradArgs a;
  a.S = S; a.L = L; a.dW = dW; a.n = n; a.Hs = Hs; a.Ws = Ws;
  a.ca = ca; a.ldS = ldS; a.coffS = coffS; a.cb = cb; a.ldL = ldL; a.coffL = coffL;
  a.relu_S = relu_S; a.scale = 1.f / net->loss_scale;
  a.scratch = net->wg_bytes > 0 ? (float*)net->shared(net->oWg) : nullptr;
  a.scratch_bytes = net->wg_bytes;
  if (!net->side) return op_wgrad(net->ctx->stream, net->dtype, a);
  // second stream: starts when everything the chain has issued so far (dz of this level) is done; the event recorded behind it
  // guards the dz buffer of this level (side_dz)
  GI_HIP(hipEventRecord(net->ev_dz, net->ctx->stream));
  GI_HIP(hipStreamWaitEvent(net->st2, net->ev_dz, 0));
  GI_TRY(op_wgrad(net->st2, net->dtype, a));
  GI_HIP(hipEventRecord(net->ev_wg[net->dz_i], net->st2));
  net->wg_busy[net->dz_i] = true;
  return GI_OK;
}

// --- second stream of a backward (gi_net::st2) ---
int side_begin(gi_net* net, int need_wgrad) {
  net->side = false;
  net->dz_i = 0;
  if (!need_wgrad || !gi_opt(GI_OPT_WGRAD_STREAM) || (net->kind == 0 && net->norm_kind != 0)) return GI_OK;
  if (!net->st2) {
    if (gi_opt(GI_OPT_WGRAD_STREAM) == 2) {   // lowest priority: the weight gradients fill what the chain leaves free
      int least = 0, greatest = 0;
      GI_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
      GI_HIP(hipStreamCreateWithPriority(&net->st2, hipStreamNonBlocking, least));
    } else {
      GI_HIP(hipStreamCreateWithFlags(&net->st2, hipStreamNonBlocking));
    }
    GI_HIP(hipEventCreateWithFlags(&net->ev_dz, hipEventDisableTiming));
    for (hipEvent_t& e : net->ev_wg) GI_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
  net->side = true;
  return GI_OK;
}
// the chain's stream waits for every weight gradient still in flight (also after an error return: the flags outlive the call)
int side_join(gi_net* net) {
  for (int i = 0; i < 3; ++i)
    if (net->wg_busy[i]) {
      net->wg_busy[i] = false;
      GI_HIP(hipStreamWaitEvent(net->ctx->stream, net->ev_wg[i], 0));
    }
  net->side = false;
  return GI_OK;
}
// the buffer the next level's dz is written to
void* side_dz(gi_net* net) {
  if (!net->side) return net->shared(net->oDr[0]);
  net->dz_i = (net->dz_i + 1) % 3;
  if (net->wg_busy[net->dz_i]) {   // read by the weight gradient three levels back
    net->wg_busy[net->dz_i] = false;
    (void)hipStreamWaitEvent(net->ctx->stream, net->ev_wg[net->dz_i], 0);
  }
  return net->shared(net->oDr[net->dz_i]);
}

int act_bn_bwd(gi_net* net, int slot, const void* g1, int ldg1, int coffg1, const void* g2, int ldg2, int coffg2,
               const void* y, int ldy, int coffy, const void* x, void* dx, int64_t pixels, int c, int act, float drop_scale,
               const BN* bn, int need_wgrad, const BwdFuse* pre = nullptr) {
  int g = (net->kind == 1 && bn) ? net->bn_groups : 1;   // BatchNorm groups: reductions per group
  int64_t pg = pixels / g;
  const int64_t T = (int64_t)net->tsz();
  // both populations in the same three launches when a reduce workgroup never straddles them
  const bool merged = g == 2 && !net->bwd_eval && pg % op_bwd_rows_per_block(pg) == 0 &&
                      (2 * (pg / op_bwd_rows_per_block(pg))) * 2 * c <= net->part_floats;
  if (merged) { g = 1; pg = pixels; }
  for (int j = 0; j < g; ++j) {
    ActBnBwdArgs a;
    memset(&a, 0, sizeof(a));
    a.groups = merged ? 2 : 1; a.stat_stride = 4 * c;
    const int64_t o = (int64_t)j * pg;
    a.g1 = g1 ? (const char*)g1 + o * ldg1 * T : nullptr; a.ldg1 = ldg1; a.coffg1 = coffg1;
    a.g2 = g2 ? (const char*)g2 + o * ldg2 * T : nullptr; a.ldg2 = ldg2; a.coffg2 = coffg2;
    a.y = (const char*)y + o * ldy * T; a.ldy = ldy; a.coffy = coffy;
    a.x = x ? (const char*)x + o * c * T : nullptr; a.dx = (char*)dx + o * c * T;
    a.pixels = pg; a.c = c; a.act = act; a.drop_scale = drop_scale;
    a.has_bn = bn ? 1 : 0;
    a.eval_bn = net->bwd_eval;
    if (bn) {
      BNPtrs p = bn_ptrs(net, slot, *bn, j);
      a.gamma = net->params + bn->gamma_off; a.save_mean = p.mean; a.save_invstd = p.inv;
      if (drop_scale == 1.f && x) { a.fwd_scale = p.scale; a.fwd_shift = p.shift; }   // dropout zeros live only in y
      a.dgamma = need_wgrad ? net->grads + bn->gamma_off : nullptr;
      a.dbeta = need_wgrad ? net->grads + bn->beta_off : nullptr;
    }
    a.inv_loss_scale = 1.f / net->loss_scale;
    a.partials = (float*)net->shared(net->oPart);
    a.sums = (float*)net->shared(net->oSums);
    if (pre && pre->planned) {   // the accumulator block was chosen before the producing GEMM ran (bwd_fuse_plan)
      GI_REQUIRE(bn && (merged || g == 1), "internal: fused BatchNorm-backward plan without the one-launch reduction");
      a.acc = pre->acc;
      a.acc_reps = pre->reps_gemm > pre->reps_reduce ? pre->reps_gemm : pre->reps_reduce;
      a.zero_next = pre->zero_next; a.zero_words = pre->zero_words;
      a.reduce_done = pre->applied ? 1 : 0;
    } else if (bn && net->use_acc && bn->acc_off >= 0 && (merged || g == 1)) {   // exact accumulators: no sums launch (stat_acc.h)
      unsigned long long* base = (unsigned long long*)net->shared(net->oAcc) + bn->acc_off + 2 * bn->acc_block();
      a.acc = base + bn->bwd_par * bn->acc_block();
      if (!net->bwd_eval) {   // (a running-statistics backward adds nothing: the block stays clean)
        a.acc_reps = stat_reps_for((pg + op_bwd_rows_per_block(pg / a.groups) - 1) / op_bwd_rows_per_block(pg / a.groups));
        const int other = 2 + (bn->bwd_par ^ 1);
        a.zero_words = (int)gi_stat_block_words(bn->c, bn->dirty[other]);
        a.zero_next = a.zero_words > 0 ? base + (bn->bwd_par ^ 1) * bn->acc_block() : nullptr;
        bn->dirty[other] = 0;
        bn->dirty[2 + bn->bwd_par] = a.acc_reps;
        bn->bwd_par ^= 1;
      }
    }
    GI_TRY(op_act_bn_bwd(net->ctx->stream, net->dtype, a));
  }
  return GI_OK;
}

__global__ void __launch_bounds__(256) fill_f32_kernel(float* dst, int count, float v) {
  for (int i = threadIdx.x; i < count; i += 256) dst[i] = v;
}

// dst += scale * sum(src), bit-reproducible: per-block sums (fp64) to part[gridDim.x], then ONE block adds them in index order
__global__ void __launch_bounds__(256) sum_part_kernel(const float* src, int64_t count, double* part) {
  __shared__ double sh[4];
  double s = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) s += src[i];
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
__global__ void __launch_bounds__(256) sum_finish_kernel(const double* part, int blocks, float* dst, float scale) {
  __shared__ double sh[256];
  sh[threadIdx.x] = threadIdx.x < blocks ? part[threadIdx.x] : 0.0;   // blocks <= 256
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) dst[0] += (float)(sh[0] * scale);
}

int grid1d(int64_t work) {
  int64_t b = (work + 255) / 256;
  if (b > 8192) b = 8192;
  return (int)(b < 1 ? 1 : b);
}
// y[n][j][p] = tanh(u[n][p][j] + bias[j]), j < out_c, u NHWC with 64 (padded) channels
template <typename T>
__global__ void __launch_bounds__(256) head_tanh_kernel(const T* __restrict__ u, const float* __restrict__ bias, float* __restrict__ y, int n,
                                                        int out_c, int hw) {
  const int64_t total = (int64_t)n * out_c * hw;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int p = (int)(i % hw);
    const int64_t t = i / hw;
    const int j = (int)(t % out_c);
    const int64_t img = t / out_c;
    y[i] = tanhf((float)u[(img * hw + p) * 64 + j] + bias[j]);
  }
}
// g (n,out_c,hw) fp32 -> NHWC T with 64 channels, zero beyond out_c
template <typename T>
__global__ void __launch_bounds__(256) pad_dy_kernel(const float* __restrict__ g, T* __restrict__ out, int n, int out_c, int hw) {
  const int64_t total = (int64_t)n * hw * 64;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int j = (int)(i & 63);
    const int64_t t = i >> 6;
    const int p = (int)(t % hw);
    const int64_t img = t / hw;
    out[i] = j < out_c ? (T)g[(img * out_c + j) * hw + p] : (T)0.f;
  }
}

// ---------------------------------------------------------------------------------------------
// Inference (gi_net_set_inference + eval mode): every BatchNorm is a fixed affine map, so its scale goes into the
// preceding convolution's weights (a second copy in the forward's layout), its shift becomes the GEMM's bias and
// the activation runs in the GEMM epilogue, which writes straight into the skip buffers: no normalisation pass at all.
// The folded copies are rebuilt when the generation counter says weights or running statistics changed.
int unet_sync_inference(gi_net* net) {
  hipStream_t st = net->ctx->stream;
  const int nd = net->nd;
  auto aff = [&](const BN& b) { return (float*)net->shared(b.inf_off); };
  PackJobs P;
  P.n = 0;
  for (int k = 2; k <= nd; ++k) {
    for (int up = 0; up < 2; ++up) {
      if (!up && k == nd) continue;   // the innermost convolution has no norm
      const BN& b = up ? net->unorm[k] : net->dnorm[k];
      const Conv& c = up ? net->up[k] : net->conv[k];
      float* a = aff(b);
      GI_TRY(op_bn_finalize(st, nullptr, 0, b.c, 1, net->params + b.gamma_off, net->params + b.beta_off, net->buffers + b.rmean_off,
                            net->buffers + b.rvar_off, a, a + b.c, a + 2 * b.c, a + 3 * b.c, 0, 0.1f, 1e-5f));
      if (P.n == 16) { GI_TRY(op_pack_weights_batch(st, net->dtype, P)); P.n = 0; }
      PackJob& J = P.j[P.n++];
      J.w = net->params + c.w_off; J.ca = c.ca; J.cb = c.cb; J.tile0 = 0;
      J.packed = up ? nullptr : (void*)net->shared(c.inf_off);
      J.phase = up ? (void*)net->shared(c.inf_off) : nullptr;
      J.scale = a; J.scale_on_b = up;
    }
  }
  GI_TRY(op_pack_weights_batch(st, net->dtype, P));
  net->inf_gen = net->affine_gen;
  return GI_OK;
}

int unet_forward_inference(gi_net* net, int s, const float* x, float* y, int n) {
  hipStream_t st = net->ctx->stream;
  const int nd = net->nd, dt = net->dtype;
  if (net->inf_gen != net->affine_gen) GI_TRY(unet_sync_inference(net));
  net->slot_n[s] = n;
  net->slot_train[s] = 0;
  net->slot_inference[s] = 1;
  net->slot_fused_u2[s] = 0;
  auto C = [&](int k) { return (void*)net->slot(s, net->oC[k]); };
  auto shift = [&](const BN& b) { return (const float*)net->shared(b.inf_off) + b.c; };
  GI_TRY(op_c1_gather(st, dt, x, net->params + net->conv[1].w_off, C(1), n, net->Hk[1], net->Wk[1], net->ch[1], 2 * net->ch[1], 0,
                      GI_ACT_LRELU, 1.f));
  for (int k = 2; k <= nd; ++k) {
    if (k < nd)
      GI_TRY(igemm(net, 0, C(k - 1), net->ch[k - 1], 2 * net->ch[k - 1], 0, net->shared(net->conv[k].inf_off), C(k), net->ch[k],
                   2 * net->ch[k], 0, n, net->Hk[k], net->Wk[k], 0, GI_ACT_LRELU, false, nullptr, 0, nullptr, 0, 0.f, nullptr, nullptr, 0,
                   shift(net->dnorm[k])));
    else
      GI_TRY(igemm(net, 0, C(k - 1), net->ch[k - 1], 2 * net->ch[k - 1], 0, packed_ptr(net, net->conv[k]), net->slot(s, net->oE),
                   net->ch[k], net->ch[k], 0, n, net->Hk[k], net->Wk[k], 0, GI_ACT_RELU, false, nullptr));
  }
  for (int k = nd; k >= 2; --k) {
    const void* in = (k == nd) ? (const void*)net->slot(s, net->oE) : C(k);
    const int cin = net->up[k].ca, co = net->ch[k - 1];
    GI_TRY(igemm(net, 1, in, cin, cin, 0, net->shared(net->up[k].inf_off), C(k - 1), co, 2 * co, co, n, net->Hk[k], net->Wk[k], k < nd ? 1 : 0,
                 GI_ACT_RELU, false, nullptr, k < nd ? net->ch[k] : 0, nullptr, 0, 0.f, nullptr, nullptr, 0, shift(net->unorm[k])));
  }
  float* osave = (float*)net->slot(s, net->oOut);
  if (net->out_c == 1) {
    GI_TRY(op_c1_scatter(st, dt, C(1), net->params + net->up[1].w_off, net->params + net->up[1].bias_off, osave, n, net->Hk[1],
                         net->Wk[1], 2 * net->ch[1], 2 * net->ch[1], 0, 1, 1, 1.f, net->shared(net->oCol), y));
    return GI_OK;
  }
  const int c1 = 2 * net->ch[1], H = net->H, W = net->W;
  if (op_c1_head4_ok(dt, c1, net->out_c, net->Wk[1], c1, 0))   // the face-parsing network's head: 64-row col GEMM + overlap-add
    return op_c1_head4_forward(st, C(1), net->params + net->up[1].w_off, net->params + net->up[1].bias_off, osave, y, n, net->Hk[1], net->Wk[1], c1,
                               0, 1, net->shared(net->oCol));
  void* U1 = net->shared(net->oU1);
  GI_TRY(igemm(net, 1, C(1), c1, c1, 0, net->shared(net->oUp1Phase), U1, 64, 64, 0, n, net->Hk[1], net->Wk[1], 1, GI_ACT_NONE, false, nullptr,
               net->ch[1]));
  const int64_t total = (int64_t)n * net->out_c * H * W;
  if (dt == GI_F16)
    hipLaunchKernelGGL(head_tanh_kernel<half_t>, dim3(grid1d(total)), dim3(256), 0, st, (const half_t*)U1, net->params + net->up[1].bias_off, y, n,
                       net->out_c, H * W);
  else
    hipLaunchKernelGGL(head_tanh_kernel<float>, dim3(grid1d(total)), dim3(256), 0, st, (const float*)U1, net->params + net->up[1].bias_off, y, n,
                       net->out_c, H * W);
  GI_LAUNCH_CHECK();
  return GI_OK;
}

// generator with InstanceNorm2d / without norm layers: the normalisation step after a convolution (raw -> dst with
// activation and dropout). InstanceNorm keeps (mean, inv) per image and channel in the slot for the backward.
int plain_norm_forward(gi_net* net, int slot, const BN& b, const void* raw, int n, int hw, void* dst, int ldy, int coffy, int act,
                       const uint8_t* drop, float drop_scale) {
  hipStream_t st = net->ctx->stream;
  if (net->norm_kind == 1)
    return op_in_forward(st, net->dtype, raw, dst, n, hw, b.c, ldy, coffy, act, drop, drop_scale, 1e-5f,
                         (float*)net->slot(slot, net->oInStats) + b.in_off);
  return op_bn_apply(st, net->dtype, raw, dst, (int64_t)n * hw, b.c, ldy, coffy, nullptr, nullptr, act, drop, drop_scale);
}

const float* conv_bias(const gi_net* net, const Conv& c) { return c.bias_off >= 0 ? net->params + c.bias_off : nullptr; }

// dbias += (1 / loss scale) * column sums of the convolution's output gradient (InstanceNorm generator: use_bias)
int bias_grad(gi_net* net, const Conv& c, const void* dz, int64_t pixels, int ch) {
  if (c.bias_off < 0) return GI_OK;
  GI_REQUIRE((pixels / 32 + 8) * 2 * ch <= net->part_floats, "internal: partials buffer too small for a bias gradient");
  return op_bias_grad(net->ctx->stream, net->dtype, dz, pixels, ch, 1.f / net->loss_scale, net->grads + c.bias_off,
                      (float*)net->shared(net->oPart));
}

int unet_forward_plain(gi_net* net, int s, const float* x, float* y, int n);
int unet_backward_plain(gi_net* net, int s, const float* dy, float* dx, int need_wgrad, int phase);

int unet_forward(gi_net* net, int s, const float* x, float* y, int n) {
  hipStream_t st = net->ctx->stream;
  const int nd = net->nd, dt = net->dtype, train = net->train;
  if (net->norm_kind != 0) return unet_forward_plain(net, s, x, y, n);
  if (!train && net->inference && nd >= 3) return unet_forward_inference(net, s, x, y, n);
  net->slot_inference[s] = 0;
  if (train) ++net->affine_gen;   // running statistics change below
  const int H = net->H, W = net->W;
  net->slot_n[s] = n;
  net->slot_train[s] = train;
  const bool fuse_u2 = nd >= 3 && net->out_c == 1 && net->fuse_head && op_c1_affine_ok(dt, 2 * net->ch[1], net->Wk[1], 2 * net->ch[1], 0);
  net->slot_fused_u2[s] = fuse_u2 ? 1 : 0;
  auto C = [&](int k) { return (void*)net->slot(s, net->oC[k]); };
  // d1: Conv2d(1->ngf) then the next block's in-place LeakyReLU (networks.py:287): the skip IS lrelu(x)
  GI_TRY(op_c1_gather(st, dt, x, net->params + net->conv[1].w_off, C(1), n, net->Hk[1], net->Wk[1], net->ch[1], 2 * net->ch[1], 0,
                      GI_ACT_LRELU, 1.f, nullptr, gi_opt(GI_OPT_MASK_BITS) ? (unsigned long long*)net->slot(s, net->oBits1) : nullptr, &net->slot_bits1[s]));
  // the copy of the input that d1's weight gradient reads (8.4 MB at the headline shape) rides in d2's normalisation pass
  // (bn_forward's side copy); where that pass does not exist, a device copy
  bool x_saved = false;
  for (int k = 2; k <= nd; ++k) {
    const int64_t pix = (int64_t)n * net->Hk[k] * net->Wk[k];
    int nt = 0;
    if (k < nd) {
      void* R = net->slot(s, net->oR[k]);
      StatPlan sp = stat_plan(net, net->dnorm[k], pix, (pix + 127) / 128, train);
      // small layers: the GEMM's last finisher per channel column normalises the column itself (IgemmFold; igemm7 decides)
      IgemmFold fold;
      int folded = 0;
      if (sp.use && dt == GI_F16) {
        fold.bn = bn_acc_args(net, s, net->dnorm[k], sp, pix, 1);
        fold.dst = C(k); fold.lddst = 2 * net->ch[k]; fold.coffdst = 0; fold.act = GI_ACT_LRELU;
        fold.drop_mask = nullptr; fold.drop_scale = 1.f; fold.drop_seed = 0; fold.drop_p = 0.f;
      }
      GI_TRY(igemm(net, 0, C(k - 1), net->ch[k - 1], 2 * net->ch[k - 1], 0, packed_ptr(net, net->conv[k]), R, net->ch[k],
                   net->ch[k], 0, n, net->Hk[k], net->Wk[k], 0, GI_ACT_NONE, true, &nt, 0, nullptr, 0, 0.f, nullptr, nullptr, 0, nullptr, &sp, nullptr,
                   (sp.use && dt == GI_F16) ? &fold : nullptr, &folded));
      if (folded) { bn_acc_commit(net, s, net->dnorm[k], sp); continue; }
      const bool side = !x_saved && k == 2 && sp.use && ((int64_t)n * H * W * 4) % 16 == 0;
      GI_TRY(bn_forward(net, s, net->dnorm[k], R, pix, nt, C(k), 2 * net->ch[k], 0, GI_ACT_LRELU, nullptr, 1.f, train, true, &sp, 0.f, 0,
                        side ? x : nullptr, side ? net->slot(s, net->oX) : nullptr, side ? (int64_t)n * H * W * 4 : 0));
      if (side) x_saved = true;
    } else {  // innermost: no down-norm; uprelu follows directly (networks.py:299-305)
      GI_TRY(igemm(net, 0, C(k - 1), net->ch[k - 1], 2 * net->ch[k - 1], 0, packed_ptr(net, net->conv[k]), net->slot(s, net->oE),
                   net->ch[k], net->ch[k], 0, n, net->Hk[k], net->Wk[k], 0, GI_ACT_RELU, false, nullptr));
    }
  }
  BnAccArgs u2_acc;
  bool u2_deferred = false;
  if (!x_saved) GI_HIP(hipMemcpyAsync(net->slot(s, net->oX), x, (size_t)n * H * W * 4, hipMemcpyDeviceToDevice, st));
  for (int k = nd; k >= 2; --k) {
    const void* in = (k == nd) ? (const void*)net->slot(s, net->oE) : C(k);
    const int cin = net->up[k].ca;
    const int64_t opix = (int64_t)n * net->Hk[k - 1] * net->Wk[k - 1];
    const int co = net->ch[k - 1];
    void* U = net->slot(s, net->oU[k]);
    int nt = 0;
    StatPlan sp = stat_plan(net, net->unorm[k], (int64_t)n * net->Hk[k] * net->Wk[k], ((int64_t)n * net->Hk[k] * net->Wk[k] + 127) / 128 * 4, train);
    const uint8_t* drop = nullptr;
    float drop_p = 0.f;
    uint64_t drop_seed = 0;
    if (train && net->dropout_p > 0.f && k >= 5 && k <= nd - 1) {
      uint8_t* m = (uint8_t*)net->slot(s, net->oMask[k]);
      if (net->ext_mask[s][k]) {
        GI_TRY(op_mask_nchw_to_nhwc(st, net->ext_mask[s][k], m, n, co, net->Hk[k - 1] * net->Wk[k - 1], 1));
      } else {
        drop_seed = net->drop_seed + 0x1000003ull * (++net->drop_counter);
      }
      drop = m;
    }
    const bool fused = (k == 2) && fuse_u2;
    IgemmFold fold;
    int folded = 0;
    const bool offer = sp.use && dt == GI_F16 && !fused;
    if (offer) {
      fold.bn = bn_acc_args(net, s, net->unorm[k], sp, opix, 1);
      fold.dst = C(k - 1); fold.lddst = 2 * co; fold.coffdst = co; fold.act = GI_ACT_RELU;
      fold.drop_mask = (uint8_t*)drop; fold.drop_scale = drop ? 1.f / (1.f - net->dropout_p) : 1.f;
      fold.drop_seed = drop_seed; fold.drop_p = (drop && !net->ext_mask[s][k]) ? net->dropout_p : 0.f;
    }
    GI_TRY(igemm(net, 1, in, cin, cin, 0, phase_ptr(net, net->up[k]), U, co, co, 0, n, net->Hk[k], net->Wk[k], k < nd ? 1 : 0,
                 GI_ACT_NONE, true, &nt, k < nd ? net->ch[k] : 0, nullptr, 0, 0.f, nullptr, nullptr, 0, nullptr, &sp, nullptr, offer ? &fold : nullptr,
                 &folded));
    if (folded) { bn_acc_commit(net, s, net->unorm[k], sp); continue; }
    if (drop && !net->ext_mask[s][k]) {
      if (sp.use) drop_p = net->dropout_p;      // drawn (same hash, same masks) inside the normalisation pass
      else GI_TRY(op_fill_dropout(st, (uint8_t*)drop, opix * co, drop_seed, net->dropout_p));
    }
    // the decoder half is only ever consumed through the parent's in-place ReLU (networks.py:289): store
    // relu(u) so that consumers need the ReLU on the skip half only; [u > 0] masks are unchanged
    // the last decoder level feeds only the single-channel head: its BatchNorm + ReLU is applied by the head's
    // kernels while they read the raw tensor (C1Affine), so the upper half of C(1) is never written
    if (fused && sp.use && net->out_c == 1) {   // the head's first kernel derives scale / shift itself (C1Affine::bn): no finalize launch
      u2_acc = bn_acc_args(net, s, net->unorm[k], sp, opix, 1);
      bn_acc_commit(net, s, net->unorm[k], sp);
      u2_deferred = true;
      continue;
    }
    GI_TRY(bn_forward(net, s, net->unorm[k], U, opix, nt, C(k - 1), 2 * co, co, GI_ACT_RELU, drop,
                      drop ? 1.f / (1.f - net->dropout_p) : 1.f, train, !fused, &sp, drop_p, drop_seed));
  }
  float* osave = (float*)net->slot(s, net->oOut);
  if (net->out_c == 1) {
    C1Affine aff;
    if (fuse_u2) {
      BNPtrs p = bn_ptrs(net, s, net->unorm[2]);
      aff.x2 = net->slot(s, net->oU[2]); aff.ld2 = net->ch[1]; aff.scale = p.scale; aff.shift = p.shift;
      if (u2_deferred) aff.bn = &u2_acc;
    }
    GI_TRY(op_c1_scatter(st, dt, C(1), net->params + net->up[1].w_off, net->params + net->up[1].bias_off, osave, n, net->Hk[1],
                         net->Wk[1], 2 * net->ch[1], 2 * net->ch[1], 0, 1, 1, 1.f, net->shared(net->oCol), y, fuse_u2 ? &aff : nullptr));
    return GI_OK;   // y written beside the saved output
  } else {
    // u1 with out_c channels: the same sub-pixel GEMM as the other up-convolutions on weights zero-padded to 64
    // output channels, then bias + tanh of the first out_c channels into the (n,out_c,H,W) fp32 result
    const int c1 = 2 * net->ch[1];
    if (op_c1_head4_ok(dt, c1, net->out_c, net->Wk[1], c1, 0))   // the face-parsing network's head: 64-row col GEMM + overlap-add
      return op_c1_head4_forward(st, C(1), net->params + net->up[1].w_off, net->params + net->up[1].bias_off, osave, y, n, net->Hk[1], net->Wk[1], c1,
                                 0, 1, net->shared(net->oCol));
    void* U1 = net->shared(net->oU1);
    GI_TRY(igemm(net, 1, C(1), c1, c1, 0, net->shared(net->oUp1Phase), U1, 64, 64, 0, n, net->Hk[1], net->Wk[1], 1, GI_ACT_NONE, false,
                 nullptr, net->ch[1]));
    const int64_t total = (int64_t)n * net->out_c * H * W;
    if (dt == GI_F16)
      hipLaunchKernelGGL(head_tanh_kernel<half_t>, dim3(grid1d(total)), dim3(256), 0, st, (const half_t*)U1, net->params + net->up[1].bias_off,
                         osave, n, net->out_c, H * W);
    else
      hipLaunchKernelGGL(head_tanh_kernel<float>, dim3(grid1d(total)), dim3(256), 0, st, (const float*)U1, net->params + net->up[1].bias_off,
                         osave, n, net->out_c, H * W);
    GI_LAUNCH_CHECK();
  }
  GI_HIP(hipMemcpyAsync(y, osave, (size_t)n * net->out_c * H * W * 4, hipMemcpyDeviceToDevice, st));
  return GI_OK;
}

// phase: 0 = whole backward; 1 = decoder half (u1 .. u_nd: their parameter gradients occupy the tail
// of the flat gradient buffer and are complete when this returns); 2 = encoder half (d_nd .. d1).
// Splitting lets the host start the decoder gradients' all-reduce while the encoder half runs.
int unet_backward(gi_net* net, int s, const float* dy, float* dx, int need_wgrad, int phase) {
  if (net->norm_kind != 0) return unet_backward_plain(net, s, dy, dx, need_wgrad, phase);
  hipStream_t st = net->ctx->stream;
  const int nd = net->nd, dt = net->dtype;
  const int H = net->H, W = net->W, n = net->slot_n[s];
  // a forward in eval mode (running-statistics BatchNorm, no dropout) can be differentiated w.r.t. its input only:
  // the frozen segmentation network of the face-parsing loss (wgan_perceptual_style_faceparsing.py:212-213)
  GI_REQUIRE(n > 0 && (net->slot_train[s] || !need_wgrad), "unet_backward: slot %d holds no train-mode forward", s);
  GI_REQUIRE(!net->slot_inference[s], "unet_backward: slot %d holds an inference forward (gi_net_set_inference): nothing was saved for a backward", s);
  GI_REQUIRE(net->out_c == 1 || !need_wgrad, "unet_backward: parameter gradients of a %d-channel generator are not built", net->out_c);
  const int evalbn = net->slot_train[s] ? 0 : 1;
  net->bwd_eval = evalbn;
  const float LS = net->loss_scale, iLS = 1.f / LS;
  auto C = [&](int k) { return (void*)net->slot(s, net->oC[k]); };
  auto gC = [&](int k) { return (void*)net->shared(net->ogC[k]); };
  auto gA = [&](int k) { return (void*)net->shared(net->ogA[k]); };
  void* D = net->shared(net->oD);   // dz of the level in progress (side_dz: rotates when the weight gradients run on the second stream)
  float* G0 = (float*)net->shared(net->oG0);
  const int64_t npx = (int64_t)n * H * W;
  if (phase == 0 || phase == 1) {
  // tanh' and loss scaling
  GI_TRY(op_tanh_bwd(st, dy, (const float*)net->slot(s, net->oOut), G0, npx * net->out_c, LS));
  // u1: ConvTranspose2d(2ngf -> 1) + bias
  const int c1 = 2 * net->ch[1];
  if (need_wgrad) {
    // (the u1 bias gradient: a sum over every output pixel, fixed order)
    hipLaunchKernelGGL(sum_part_kernel, dim3(256), dim3(256), 0, st, G0, npx, (double*)net->shared(net->oPart));
    hipLaunchKernelGGL(sum_finish_kernel, dim3(1), dim3(256), 0, st, (const double*)net->shared(net->oPart), 256, net->grads + net->up[1].bias_off, iLS);
    GI_LAUNCH_CHECK();
    C1Affine aff;
    const bool fused = net->slot_fused_u2[s] != 0;
    if (fused) {   // the upper half of C(1) was never written (unet_forward): recompute it from the raw decoder output
      BNPtrs p = bn_ptrs(net, s, net->unorm[2]);
      aff.x2 = net->slot(s, net->oU[2]); aff.ld2 = net->ch[1]; aff.scale = p.scale; aff.shift = p.shift;
    }
    GI_TRY(op_c1_wgrad(st, dt, C(1), G0, net->grads + net->up[1].w_off, n, net->Hk[1], net->Wk[1], c1, c1, 0, 1, iLS, 1.f, fused ? &aff : nullptr,
                       (float*)net->shared(net->oPart), net->part_floats));
  }
  if (net->out_c == 1) {
    GI_TRY(op_c1_gather(st, dt, G0, net->params + net->up[1].w_off, gC(1), n, net->Hk[1], net->Wk[1], c1, c1, 0, GI_ACT_NONE, 1.f));
  } else if (op_c1_head4_ok(dt, c1, net->out_c, net->Wk[1], c1, 0)) {
    GI_TRY(op_c1_head4_dgrad(st, G0, net->params + net->up[1].w_off, gC(1), n, net->Hk[1], net->Wk[1], c1, 0));
  } else {
    void* U1 = net->shared(net->oU1);
    if (dt == GI_F16) hipLaunchKernelGGL(pad_dy_kernel<half_t>, dim3(grid1d(npx * 64)), dim3(256), 0, st, G0, (half_t*)U1, n, net->out_c, H * W);
    else hipLaunchKernelGGL(pad_dy_kernel<float>, dim3(grid1d(npx * 64)), dim3(256), 0, st, G0, (float*)U1, n, net->out_c, H * W);
    GI_LAUNCH_CHECK();
    const void* wp = net->oUp1Packed >= 0 ? (const void*)net->shared(net->oUp1Packed) : (const void*)net->shared(net->oUp1Pad);
    GI_TRY(igemm(net, 0, U1, 64, 64, 0, wp, gC(1), c1, c1, 0, n, net->Hk[1], net->Wk[1], 0, GI_ACT_NONE, false, nullptr));
  }
  // decoder, outermost -> innermost: level k's concat gradient feeds up[k+1]
  // a reduction planned by an earlier, abandoned backward left sums in its accumulator block
  for (int k = 2; k <= nd; ++k) {
    if (net->unorm[k].c == 0 || net->unorm[k].id < 0) continue;
    BwdFuse& pend = net->bwd_pending[net->unorm[k].id];
    if (pend.planned) GI_HIP(hipMemsetAsync(pend.acc, 0, (size_t)net->unorm[k].acc_block() * 8, st));
    pend = BwdFuse();
  }
  for (int k = 1; k <= nd - 1; ++k) {
    const int kk = k + 1;
    const int ck = net->ch[k];
    const int64_t pix = (int64_t)n * net->Hk[k] * net->Wk[k];
    const float ds = (!evalbn && net->dropout_p > 0.f && kk >= 5 && kk <= nd - 1) ? 1.f / (1.f - net->dropout_p) : 1.f;
    D = side_dz(net);
    BwdFuse& pend = net->bwd_pending[net->unorm[kk].id];
    GI_TRY(act_bn_bwd(net, s, nullptr, 0, 0, gC(k), 2 * ck, ck, C(k), 2 * ck, ck, net->slot(s, net->oU[kk]), D, pix, ck,
                      GI_ACT_NONE, ds, &net->unorm[kk], need_wgrad, pend.planned ? &pend : nullptr));
    pend = BwdFuse();
    const void* Sin = (kk == nd) ? (const void*)net->slot(s, net->oE) : C(kk);
    const int ca = net->up[kk].ca;
    if (need_wgrad)
      GI_TRY(wgrad(net, Sin, ca, ca, 0, kk < nd ? 1 : 0, D, ck, ck, 0, n, net->Hk[kk], net->Wk[kk], net->grads + net->up[kk].w_off));
    void* gout = (kk == nd) ? (void*)net->shared(net->ogE) : gC(kk);
    // the upper half of this GEMM's columns is the gradient at the output of the NEXT decoder level's BatchNorm (its ReLU is the
    // consumer's): that layer's backward reduction rides in the epilogue (IgemmArgs::bwd_c0; levels without dropout, kernels that take it)
    BwdFuse* pbf = nullptr;
    const int kn = kk + 1;
    if (kk <= nd - 1 && !(net->dropout_p > 0.f && !evalbn && kn >= 5 && kn <= nd - 1)) {
      const int64_t pixn = (int64_t)n * net->Hk[kk] * net->Wk[kk];
      pbf = &net->bwd_pending[net->unorm[kn].id];
      *pbf = bwd_fuse_plan(net, s, net->unorm[kn], net->slot(s, net->oU[kn]), pixn, pixn / 256, 0.f);
      pbf->c0 = ca / 2; pbf->c = ca / 2;
    }
    GI_TRY(igemm(net, 0, D, ck, ck, 0, packed_ptr(net, net->up[kk]), gout, ca, ca, 0, n, net->Hk[kk], net->Wk[kk], 0, GI_ACT_NONE,
                 false, nullptr, 0, nullptr, 0, 0.f, nullptr, nullptr, 0, nullptr, nullptr, pbf));
  }
  }  // decoder half
  if (phase == 1) return GI_OK;
  // the encoder half may be split once more: phase 3 = innermost .. level 5 (their parameter gradients, the bulk
  // of the encoder's, are complete first), phase 4 = levels 4 .. 1; phase 2 (or 0) runs both
  const bool run_inner = phase != 4, run_outer = phase != 3;
  int lrelu1_done = 0;
  C1WFuse c1w;
  // innermost conv (no norm): dz = gE * [E > 0]
  if (run_inner) {
    const int c = net->ch[nd];
    const int64_t pix = (int64_t)n * net->Hk[nd] * net->Wk[nd];
    D = side_dz(net);
    GI_TRY(act_bn_bwd(net, s, nullptr, 0, 0, net->shared(net->ogE), c, 0, net->slot(s, net->oE), c, 0, nullptr, D, pix, c, GI_ACT_NONE,
                      1.f, nullptr, need_wgrad));
    const int cb = net->ch[nd - 1];
    if (need_wgrad)
      GI_TRY(wgrad(net, D, c, c, 0, 0, C(nd - 1), cb, 2 * cb, 0, n, net->Hk[nd], net->Wk[nd], net->grads + net->conv[nd].w_off));
    GI_TRY(igemm(net, 1, D, c, c, 0, phase_ptr(net, net->conv[nd]), gA(nd - 1), cb, cb, 0, n, net->Hk[nd], net->Wk[nd], 0, GI_ACT_NONE,
                 false, nullptr));
  }
  // encoder, innermost -> outermost
  for (int k = nd - 1; k >= 2; --k) {
    if (k >= 5 ? !run_inner : !run_outer) continue;
    const int c = net->ch[k];
    const int64_t pix = (int64_t)n * net->Hk[k] * net->Wk[k];
    D = side_dz(net);
    GI_TRY(act_bn_bwd(net, s, gA(k), c, 0, gC(k), 2 * c, 0, C(k), 2 * c, 0, net->slot(s, net->oR[k]), D, pix, c, GI_ACT_LRELU, 1.f,
                      &net->dnorm[k], need_wgrad));
    const int cb = net->ch[k - 1];
    if (need_wgrad)
      GI_TRY(wgrad(net, D, c, c, 0, 0, C(k - 1), cb, 2 * cb, 0, n, net->Hk[k], net->Wk[k], net->grads + net->conv[k].w_off));
    // d1 has no norm: its backward, (g + [y>0] * skip gradient) * LeakyReLU'(y), rides in the epilogue of d2's
    // input-gradient GEMM when the kernel supports it (same arithmetic, one 4-tensor HBM pass less)
    const bool fuse1 = (k == 2);
    // ... and d1's weight gradient is formed from that GEMM's tiles while they are in LDS (IgemmArgs::c1w_*): the gradient at d1's
    // output is never stored (the generator's input gradient is not asked for) and never read back
    C1WFuse* pc1w = nullptr;
    if (fuse1 && need_wgrad && net->oC1w >= 0 && gi_opt(GI_OPT_C1W_FUSE)) {
      c1w.img = (const float*)net->slot(s, net->oX); c1w.part = (float*)net->shared(net->oC1w); c1w.part_floats = net->c1w_floats - 64 * 1024;
      c1w.scale = iLS; c1w.skip_out = dx ? 0 : 1;
      pc1w = &c1w;
    }
    GI_TRY(igemm(net, 1, D, c, c, 0, phase_ptr(net, net->conv[k]), gA(k - 1), cb, cb, 0, n, net->Hk[k], net->Wk[k], 0, GI_ACT_NONE,
                 false, nullptr, 0, fuse1 ? C(1) : nullptr, 2 * cb, 0.2f, fuse1 ? &lrelu1_done : nullptr, fuse1 ? gC(1) : nullptr, 2 * cb, nullptr, nullptr,
                 nullptr, nullptr, nullptr, (fuse1 && net->slot_bits1[s] && gi_opt(GI_OPT_MASK_BITS)) ? (const unsigned long long*)net->slot(s, net->oBits1) : nullptr,
                 pc1w));
  }
  if (run_outer) {
    const int c = net->ch[1];
    const int64_t pix = (int64_t)n * net->Hk[1] * net->Wk[1];
    void* D1 = D = side_dz(net);
    if (lrelu1_done) D1 = gA(1);
    else GI_TRY(act_bn_bwd(net, s, gA(1), c, 0, gC(1), 2 * c, 0, C(1), 2 * c, 0, nullptr, D, pix, c, GI_ACT_LRELU, 1.f, nullptr, need_wgrad));
    if (need_wgrad && c1w.applied)
      GI_TRY(op_c1_wgrad_reduce(st, c1w.part, net->grads + net->conv[1].w_off, c * 16, c1w.blocks, c1w.part + (int64_t)c1w.blocks * 1024,
                                net->c1w_floats - (int64_t)c1w.blocks * 1024));
    else if (need_wgrad)
      GI_TRY(op_c1_wgrad(st, dt, D1, (const float*)net->slot(s, net->oX), net->grads + net->conv[1].w_off, n, net->Hk[1], net->Wk[1], c, c,
                         0, 0, iLS, 1.f, nullptr, (float*)net->shared(net->oPart), net->part_floats));
    if (dx) GI_TRY(op_c1_scatter(st, dt, D1, net->params + net->conv[1].w_off, nullptr, dx, n, net->Hk[1], net->Wk[1], c, c, 0, 0, 0, iLS,
                                 net->shared(net->oCol)));
  }
  return GI_OK;
}

// ---------------------------------------------------------------------------------------------
// Generator built with get_norm_layer('instance') or ('none') (networks.py:29-45, 270-318). Same skeleton as above on
// the unfused building blocks: convolution (+ bias) -> raw tensor -> normalisation pass (InstanceNorm statistics per
// image, or identity) with the activation and dropout. InstanceNorm has no running statistics: train and eval differ
// only by dropout. A bias in front of an InstanceNorm cancels in the forward and has a zero gradient up to rounding; it
// is still applied and differentiated like the reference does.
int unet_forward_plain(gi_net* net, int s, const float* x, float* y, int n) {
  hipStream_t st = net->ctx->stream;
  const int nd = net->nd, dt = net->dtype, train = net->train;
  const int H = net->H, W = net->W;
  net->slot_inference[s] = 0;
  net->slot_n[s] = n;
  net->slot_train[s] = train;
  net->slot_fused_u2[s] = 0;
  auto C = [&](int k) { return (void*)net->slot(s, net->oC[k]); };
  GI_HIP(hipMemcpyAsync(net->slot(s, net->oX), x, (size_t)n * H * W * 4, hipMemcpyDeviceToDevice, st));
  GI_TRY(op_c1_gather(st, dt, x, net->params + net->conv[1].w_off, C(1), n, net->Hk[1], net->Wk[1], net->ch[1], 2 * net->ch[1], 0,
                      GI_ACT_LRELU, 1.f, conv_bias(net, net->conv[1])));
  for (int k = 2; k <= nd; ++k) {
    if (k < nd) {
      void* R = net->slot(s, net->oR[k]);
      GI_TRY(igemm(net, 0, C(k - 1), net->ch[k - 1], 2 * net->ch[k - 1], 0, packed_ptr(net, net->conv[k]), R, net->ch[k], net->ch[k], 0, n,
                   net->Hk[k], net->Wk[k], 0, GI_ACT_NONE, false, nullptr, 0, nullptr, 0, 0.f, nullptr, nullptr, 0, conv_bias(net, net->conv[k])));
      GI_TRY(plain_norm_forward(net, s, net->dnorm[k], R, n, net->Hk[k] * net->Wk[k], C(k), 2 * net->ch[k], 0, GI_ACT_LRELU, nullptr, 1.f));
    } else {
      GI_TRY(igemm(net, 0, C(k - 1), net->ch[k - 1], 2 * net->ch[k - 1], 0, packed_ptr(net, net->conv[k]), net->slot(s, net->oE), net->ch[k],
                   net->ch[k], 0, n, net->Hk[k], net->Wk[k], 0, GI_ACT_RELU, false, nullptr, 0, nullptr, 0, 0.f, nullptr, nullptr, 0,
                   conv_bias(net, net->conv[k])));
    }
  }
  for (int k = nd; k >= 2; --k) {
    const void* in = (k == nd) ? (const void*)net->slot(s, net->oE) : C(k);
    const int cin = net->up[k].ca, co = net->ch[k - 1];
    const int64_t opix = (int64_t)n * net->Hk[k - 1] * net->Wk[k - 1];
    void* U = net->slot(s, net->oU[k]);
    const uint8_t* drop = nullptr;
    if (train && net->dropout_p > 0.f && k >= 5 && k <= nd - 1) {
      uint8_t* m = (uint8_t*)net->slot(s, net->oMask[k]);
      if (net->ext_mask[s][k]) GI_TRY(op_mask_nchw_to_nhwc(st, net->ext_mask[s][k], m, n, co, net->Hk[k - 1] * net->Wk[k - 1], 1));
      else GI_TRY(op_fill_dropout(st, m, opix * co, net->drop_seed + 0x1000003ull * (++net->drop_counter), net->dropout_p));
      drop = m;
    }
    GI_TRY(igemm(net, 1, in, cin, cin, 0, phase_ptr(net, net->up[k]), U, co, co, 0, n, net->Hk[k], net->Wk[k], k < nd ? 1 : 0, GI_ACT_NONE,
                 false, nullptr, k < nd ? net->ch[k] : 0, nullptr, 0, 0.f, nullptr, nullptr, 0, conv_bias(net, net->up[k])));
    GI_TRY(plain_norm_forward(net, s, net->unorm[k], U, n, net->Hk[k - 1] * net->Wk[k - 1], C(k - 1), 2 * co, co, GI_ACT_RELU, drop,
                              drop ? 1.f / (1.f - net->dropout_p) : 1.f));
  }
  float* osave = (float*)net->slot(s, net->oOut);
  if (net->out_c == 1)
    return op_c1_scatter(st, dt, C(1), net->params + net->up[1].w_off, net->params + net->up[1].bias_off, osave, n, net->Hk[1], net->Wk[1],
                         2 * net->ch[1], 2 * net->ch[1], 0, 1, 1, 1.f, net->shared(net->oCol), y);
  const int c1 = 2 * net->ch[1];
  if (op_c1_head4_ok(dt, c1, net->out_c, net->Wk[1], c1, 0))   // the face-parsing network's head: 64-row col GEMM + overlap-add
    return op_c1_head4_forward(st, C(1), net->params + net->up[1].w_off, net->params + net->up[1].bias_off, osave, y, n, net->Hk[1], net->Wk[1], c1,
                               0, 1, net->shared(net->oCol));
  void* U1 = net->shared(net->oU1);
  GI_TRY(igemm(net, 1, C(1), c1, c1, 0, net->shared(net->oUp1Phase), U1, 64, 64, 0, n, net->Hk[1], net->Wk[1], 1, GI_ACT_NONE, false, nullptr,
               net->ch[1]));
  const int64_t total = (int64_t)n * net->out_c * H * W;
  if (dt == GI_F16)
    hipLaunchKernelGGL(head_tanh_kernel<half_t>, dim3(grid1d(total)), dim3(256), 0, st, (const half_t*)U1, net->params + net->up[1].bias_off, osave,
                       n, net->out_c, H * W);
  else
    hipLaunchKernelGGL(head_tanh_kernel<float>, dim3(grid1d(total)), dim3(256), 0, st, (const float*)U1, net->params + net->up[1].bias_off, osave, n,
                       net->out_c, H * W);
  GI_LAUNCH_CHECK();
  GI_HIP(hipMemcpyAsync(y, osave, (size_t)n * net->out_c * H * W * 4, hipMemcpyDeviceToDevice, st));
  return GI_OK;
}

// backward of one normalisation step of the plain generator: dz of [dropout] -> activation, then through InstanceNorm
int plain_norm_backward(gi_net* net, int s, const BN& b, const void* g1, int ldg1, const void* g2, int ldg2, int coffg2, const void* y, int ldy,
                        int coffy, const void* x, void* dx, int n, int hw, int act, float drop_scale) {
  if (net->norm_kind != 1)
    return act_bn_bwd(net, s, g1, ldg1, 0, g2, ldg2, coffg2, y, ldy, coffy, nullptr, dx, (int64_t)n * hw, b.c, act, drop_scale, nullptr, 0);
  ActBnBwdArgs a;
  memset(&a, 0, sizeof(a));
  a.g1 = g1; a.ldg1 = ldg1; a.g2 = g2; a.ldg2 = ldg2; a.coffg2 = coffg2;
  a.y = y; a.ldy = ldy; a.coffy = coffy; a.x = x; a.dx = dx;
  a.pixels = (int64_t)n * hw; a.c = b.c; a.act = act; a.drop_scale = drop_scale;
  return op_act_in_bwd(net->ctx->stream, net->dtype, a, n, hw, (const float*)net->slot(s, net->oInStats) + b.in_off);
}

int unet_backward_plain(gi_net* net, int s, const float* dy, float* dx, int need_wgrad, int phase) {
  hipStream_t st = net->ctx->stream;
  const int nd = net->nd, dt = net->dtype;
  const int H = net->H, W = net->W, n = net->slot_n[s];
  GI_REQUIRE(n > 0, "unet_backward: slot %d holds no forward", s);
  GI_REQUIRE(net->out_c == 1 || !need_wgrad, "unet_backward: parameter gradients of a %d-channel generator are not built", net->out_c);
  net->bwd_eval = 0;
  const float LS = net->loss_scale, iLS = 1.f / LS;
  auto C = [&](int k) { return (void*)net->slot(s, net->oC[k]); };
  auto gC = [&](int k) { return (void*)net->shared(net->ogC[k]); };
  auto gA = [&](int k) { return (void*)net->shared(net->ogA[k]); };
  void* D = net->shared(net->oD);
  float* G0 = (float*)net->shared(net->oG0);
  const int64_t npx = (int64_t)n * H * W;
  const bool dropped = net->slot_train[s] && net->dropout_p > 0.f;
  if (phase == 0 || phase == 1) {
    GI_TRY(op_tanh_bwd(st, dy, (const float*)net->slot(s, net->oOut), G0, npx * net->out_c, LS));
    const int c1 = 2 * net->ch[1];
    if (need_wgrad) {
      hipLaunchKernelGGL(sum_part_kernel, dim3(256), dim3(256), 0, st, G0, npx, (double*)net->shared(net->oPart));
      hipLaunchKernelGGL(sum_finish_kernel, dim3(1), dim3(256), 0, st, (const double*)net->shared(net->oPart), 256, net->grads + net->up[1].bias_off, iLS);
      GI_LAUNCH_CHECK();
      GI_TRY(op_c1_wgrad(st, dt, C(1), G0, net->grads + net->up[1].w_off, n, net->Hk[1], net->Wk[1], c1, c1, 0, 1, iLS, 1.f, nullptr, (float*)net->shared(net->oPart), net->part_floats));
    }
    if (net->out_c == 1) {
      GI_TRY(op_c1_gather(st, dt, G0, net->params + net->up[1].w_off, gC(1), n, net->Hk[1], net->Wk[1], c1, c1, 0, GI_ACT_NONE, 1.f));
    } else if (op_c1_head4_ok(dt, c1, net->out_c, net->Wk[1], c1, 0)) {
      GI_TRY(op_c1_head4_dgrad(st, G0, net->params + net->up[1].w_off, gC(1), n, net->Hk[1], net->Wk[1], c1, 0));
    } else {
      void* U1 = net->shared(net->oU1);
      if (dt == GI_F16) hipLaunchKernelGGL(pad_dy_kernel<half_t>, dim3(grid1d(npx * 64)), dim3(256), 0, st, G0, (half_t*)U1, n, net->out_c, H * W);
      else hipLaunchKernelGGL(pad_dy_kernel<float>, dim3(grid1d(npx * 64)), dim3(256), 0, st, G0, (float*)U1, n, net->out_c, H * W);
      GI_LAUNCH_CHECK();
      const void* wp = net->oUp1Packed >= 0 ? (const void*)net->shared(net->oUp1Packed) : (const void*)net->shared(net->oUp1Pad);
      GI_TRY(igemm(net, 0, U1, 64, 64, 0, wp, gC(1), c1, c1, 0, n, net->Hk[1], net->Wk[1], 0, GI_ACT_NONE, false, nullptr));
    }
    for (int k = 1; k <= nd - 1; ++k) {
      const int kk = k + 1, ck = net->ch[k];
      const int64_t pix = (int64_t)n * net->Hk[k] * net->Wk[k];
      const float ds = (dropped && kk >= 5 && kk <= nd - 1) ? 1.f / (1.f - net->dropout_p) : 1.f;
      GI_TRY(plain_norm_backward(net, s, net->unorm[kk], nullptr, 0, gC(k), 2 * ck, ck, C(k), 2 * ck, ck, net->slot(s, net->oU[kk]), D, n,
                                 net->Hk[k] * net->Wk[k], GI_ACT_NONE, ds));
      const void* Sin = (kk == nd) ? (const void*)net->slot(s, net->oE) : C(kk);
      const int ca = net->up[kk].ca;
      if (need_wgrad) {
        GI_TRY(bias_grad(net, net->up[kk], D, pix, ck));
        GI_TRY(wgrad(net, Sin, ca, ca, 0, kk < nd ? 1 : 0, D, ck, ck, 0, n, net->Hk[kk], net->Wk[kk], net->grads + net->up[kk].w_off));
      }
      void* gout = (kk == nd) ? (void*)net->shared(net->ogE) : gC(kk);
      GI_TRY(igemm(net, 0, D, ck, ck, 0, packed_ptr(net, net->up[kk]), gout, ca, ca, 0, n, net->Hk[kk], net->Wk[kk], 0, GI_ACT_NONE, false, nullptr));
    }
  }
  if (phase == 1) return GI_OK;
  const bool run_inner = phase != 4, run_outer = phase != 3;
  if (run_inner) {
    const int c = net->ch[nd];
    const int64_t pix = (int64_t)n * net->Hk[nd] * net->Wk[nd];
    GI_TRY(act_bn_bwd(net, s, nullptr, 0, 0, net->shared(net->ogE), c, 0, net->slot(s, net->oE), c, 0, nullptr, D, pix, c, GI_ACT_NONE, 1.f,
                      nullptr, need_wgrad));
    const int cb = net->ch[nd - 1];
    if (need_wgrad) {
      GI_TRY(bias_grad(net, net->conv[nd], D, pix, c));
      GI_TRY(wgrad(net, D, c, c, 0, 0, C(nd - 1), cb, 2 * cb, 0, n, net->Hk[nd], net->Wk[nd], net->grads + net->conv[nd].w_off));
    }
    GI_TRY(igemm(net, 1, D, c, c, 0, phase_ptr(net, net->conv[nd]), gA(nd - 1), cb, cb, 0, n, net->Hk[nd], net->Wk[nd], 0, GI_ACT_NONE, false, nullptr));
  }
  for (int k = nd - 1; k >= 2; --k) {
    if (k >= 5 ? !run_inner : !run_outer) continue;
    const int c = net->ch[k];
    const int64_t pix = (int64_t)n * net->Hk[k] * net->Wk[k];
    GI_TRY(plain_norm_backward(net, s, net->dnorm[k], gA(k), c, gC(k), 2 * c, 0, C(k), 2 * c, 0, net->slot(s, net->oR[k]), D, n, net->Hk[k] * net->Wk[k],
                               GI_ACT_LRELU, 1.f));
    const int cb = net->ch[k - 1];
    if (need_wgrad) {
      GI_TRY(bias_grad(net, net->conv[k], D, pix, c));
      GI_TRY(wgrad(net, D, c, c, 0, 0, C(k - 1), cb, 2 * cb, 0, n, net->Hk[k], net->Wk[k], net->grads + net->conv[k].w_off));
    }
    GI_TRY(igemm(net, 1, D, c, c, 0, phase_ptr(net, net->conv[k]), gA(k - 1), cb, cb, 0, n, net->Hk[k], net->Wk[k], 0, GI_ACT_NONE, false, nullptr));
  }
  if (run_outer) {
    const int c = net->ch[1];
    const int64_t pix = (int64_t)n * net->Hk[1] * net->Wk[1];
    GI_TRY(act_bn_bwd(net, s, gA(1), c, 0, gC(1), 2 * c, 0, C(1), 2 * c, 0, nullptr, D, pix, c, GI_ACT_LRELU, 1.f, nullptr, need_wgrad));
    if (need_wgrad) {
      GI_TRY(bias_grad(net, net->conv[1], D, pix, c));
      GI_TRY(op_c1_wgrad(st, dt, D, (const float*)net->slot(s, net->oX), net->grads + net->conv[1].w_off, n, net->Hk[1], net->Wk[1], c, c, 0, 0, iLS, 1.f, nullptr, (float*)net->shared(net->oPart), net->part_floats));
    }
    if (dx) GI_TRY(op_c1_scatter(st, dt, D, net->params + net->conv[1].w_off, nullptr, dx, n, net->Hk[1], net->Wk[1], c, c, 0, 0, 0, iLS, net->shared(net->oCol)));
  }
  return GI_OK;
}

// ---------------------------------------------------------------------------------------------
int patchgan_forward(gi_net* net, int s, const float* x, float* y, int n) {
  hipStream_t st = net->ctx->stream;
  const int dt = net->dtype, train = net->train, H = net->H, W = net->W;
  if (train) ++net->affine_gen;   // running statistics change below
  GI_REQUIRE(n % net->bn_groups == 0, "patchgan_forward: %d images do not split into %d BatchNorm groups", n, net->bn_groups);
  net->slot_groups[s] = net->bn_groups;
  net->slot_n[s] = n;
  net->slot_train[s] = train;
  const bool fuse_a4 = net->fuse_head && op_head_affine_ok(dt, 512);
  net->slot_fused_u2[s] = fuse_a4 ? 1 : 0;   // (critic: "conv4's activation was not materialised")
  GI_TRY(op_c1_gather(st, dt, x, net->params + net->dconv[1].w_off, net->slot(s, net->oA[1]), n, H / 2, W / 2, 64, 64, 0, GI_ACT_LRELU, 1.f, nullptr,
                      gi_opt(GI_OPT_MASK_BITS) ? (unsigned long long*)net->slot(s, net->oBits1) : nullptr, &net->slot_bits1[s]));
  // the copy of the input that conv1's weight gradient reads rides in conv2's normalisation pass where that pass takes a side
  // copy (accumulator path); otherwise a device copy
  bool x_saved = false, a4_deferred = false;
  BnAccArgs a4_acc;
  for (int i = 2; i <= 4; ++i) {
    const Conv& c = net->dconv[i];
    const int Hs = H >> i, Ws = W >> i;
    int nt = 0;
    void* R = net->slot(s, net->oRd[i]);
    StatPlan sp = stat_plan(net, net->dbn[i], (int64_t)n * Hs * Ws, ((int64_t)n * Hs * Ws + 127) / 128, train);
    GI_TRY(igemm(net, 0, net->slot(s, net->oA[i - 1]), c.cb, c.cb, 0, packed_ptr(net, c), R, c.ca, c.ca, 0, n, Hs, Ws, 0, GI_ACT_NONE,
                 true, &nt, 0, nullptr, 0, 0.f, nullptr, nullptr, 0, nullptr, &sp));
    const bool side = i == 2 && sp.use && ((int64_t)n * H * W * 4) % 16 == 0;
    // conv4 feeds only the head: its BatchNorm + LeakyReLU is applied by the head kernels from the raw tensor
    if (i == 4 && fuse_a4 && sp.use && gi_opt(GI_OPT_HEAD_FAST)) {   // the head's first kernel derives conv4's scale / shift itself (HeadArgs::bn)
      a4_acc = bn_acc_args(net, s, net->dbn[i], sp, (int64_t)n * Hs * Ws / net->bn_groups, net->bn_groups);
      bn_acc_commit(net, s, net->dbn[i], sp);
      a4_deferred = true;
      continue;
    }
    GI_TRY(bn_forward(net, s, net->dbn[i], R, (int64_t)n * Hs * Ws, nt, net->slot(s, net->oA[i]), c.ca, 0, GI_ACT_LRELU, nullptr, 1.f, train,
                      !(i == 4 && fuse_a4), &sp, 0.f, 0, side ? x : nullptr, side ? net->slot(s, net->oX) : nullptr,
                      side ? (int64_t)n * H * W * 4 : 0));
    if (side) x_saved = true;
  }
  if (!x_saved) GI_HIP(hipMemcpyAsync(net->slot(s, net->oX), x, (size_t)n * H * W * 4, hipMemcpyDeviceToDevice, st));
  HeadArgs h;
  h.a4 = net->slot(s, fuse_a4 ? net->oRd[4] : net->oA[4]);
  if (fuse_a4) {
    BNPtrs p = bn_ptrs(net, s, net->dbn[4]);
    h.scale4 = p.scale; h.shift4 = p.shift; h.n_per_group = n / net->bn_groups; h.gstride = 4 * 512;
    if (a4_deferred) h.bn = &a4_acc;
  }
  if (net->oHt >= 0) { h.tbuf = (float*)net->shared(net->oHt); h.tbuf_bytes = net->ht_bytes; }
  h.w5 = net->params + net->w5_off; h.wl = net->params + net->wl_off; h.bl = net->params + net->bl_off;
  h.h = (float*)net->slot(s, net->oHh); h.out = (float*)net->slot(s, net->oOut);
  h.n = n; h.Hh = net->Hh; h.Wh = net->Wh; h.c = 512; h.sigmoid = net->sigmoid;
  h.out2 = y;
  GI_TRY(op_head_forward(st, dt, h));
  return GI_OK;
}

// phase: 0 = whole backward; 1 = head + conv4 block (their gradients, 8.5 of the critic's 11 MB, are the tail of the
// flat buffer and complete when this returns); 2 = conv3 .. conv1
int patchgan_backward(gi_net* net, int s, const float* dy, float* dx, int need_wgrad, int phase = 0) {
  hipStream_t st = net->ctx->stream;
  const int dt = net->dtype, H = net->H, W = net->W, n = net->slot_n[s];
  GI_REQUIRE(n > 0 && net->slot_train[s], "patchgan_backward: slot %d holds no train-mode forward", s);
  GI_REQUIRE(net->slot_groups[s] == net->bn_groups, "patchgan_backward: slot %d was produced with %d BatchNorm groups, the net is set to %d", s,
             net->slot_groups[s], net->bn_groups);
  net->bwd_eval = 0;
  const float LS = net->loss_scale, iLS = 1.f / LS;
  void* D = net->shared(net->oD);
  HeadBwdArgs hb;
  hb.a4 = net->slot(s, net->oA[4]); hb.w5 = net->params + net->w5_off; hb.wl = net->params + net->wl_off;
  hb.h = (const float*)net->slot(s, net->oHh); hb.out = (const float*)net->slot(s, net->oOut);
  hb.dy = dy; hb.da4 = net->shared(net->ogA[4]);
  hb.dw5 = need_wgrad ? net->grads + net->w5_off : nullptr;
  hb.dwl = need_wgrad ? net->grads + net->wl_off : nullptr;
  hb.dbl = need_wgrad ? net->grads + net->bl_off : nullptr;
  hb.dh = (float*)net->shared(net->oDh);
  hb.scratch = (float*)net->shared(net->oHw); hb.scratch_bytes = net->hw_bytes;
  if (net->slot_fused_u2[s]) {   // the forward left conv4's activation to the head kernels: so does the backward
    BNPtrs p = bn_ptrs(net, s, net->dbn[4]);
    hb.a4 = net->slot(s, net->oRd[4]);
    hb.scale4 = p.scale; hb.shift4 = p.shift; hb.n_per_group = n / net->bn_groups; hb.gstride = 4 * 512;
  }
  hb.n = n; hb.Hh = net->Hh; hb.Wh = net->Wh; hb.c = 512; hb.sigmoid = net->sigmoid; hb.loss_scale = LS;
  if (phase != 2) {
    // a reduction planned by an earlier, abandoned backward (phase 1 without its phase 2) left sums in its accumulator block
    for (int i = 2; i <= 4; ++i) {
      BwdFuse& pend = net->bwd_pending[net->dbn[i].id];
      if (pend.planned) GI_HIP(hipMemsetAsync(pend.acc, 0, (size_t)net->dbn[i].acc_block() * 8, st));
      pend = BwdFuse();
    }
    // conv4's BatchNorm-backward sums ride in the head's input-gradient blocks (HeadBwdArgs::bwd_*): no reduce launch
    BwdFuse& p4 = net->bwd_pending[net->dbn[4].id];
    int head_bw = 0;
    p4 = bwd_fuse_plan(net, s, net->dbn[4], net->slot(s, net->oRd[4]), (int64_t)n * net->Hh * net->Wh, (int64_t)n * 4, 0.2f);
    if (p4.planned) {
      hb.bwd_x = p4.x; hb.bwd_scale = p4.scale; hb.bwd_shift = p4.shift; hb.bwd_mean = p4.mean; hb.bwd_inv = p4.inv;
      hb.bwd_stride = p4.stride; hb.bwd_n_per_group = n / (p4.groups > 0 ? p4.groups : 1); hb.bwd_reps = p4.reps_gemm; hb.bwd_slope = p4.slope;
      hb.bwd_acc = p4.acc; hb.bwd_applied = &head_bw;
    }
    GI_TRY(op_head_backward(st, dt, hb));
    p4.applied = head_bw != 0;
  }
  int lrelu1_done = 0;
  C1WFuse c1w;
  for (int i = 4; i >= 2; --i) {
    if (i == 4 ? phase == 2 : phase == 1) continue;
    const Conv& c = net->dconv[i];
    const int Hs = H >> i, Ws = W >> i;
    const int64_t pix = (int64_t)n * Hs * Ws;
    BwdFuse& pend = net->bwd_pending[net->dbn[i].id];
    D = side_dz(net);
    GI_TRY(act_bn_bwd(net, s, net->shared(net->ogA[i]), c.ca, 0, nullptr, 0, 0, net->slot(s, net->oA[i]), c.ca, 0, net->slot(s, net->oRd[i]),
                      D, pix, c.ca, GI_ACT_LRELU, 1.f, &net->dbn[i], need_wgrad, pend.planned ? &pend : nullptr));
    pend = BwdFuse();
    if (need_wgrad)
      GI_TRY(wgrad(net, D, c.ca, c.ca, 0, 0, net->slot(s, net->oA[i - 1]), c.cb, c.cb, 0, n, Hs, Ws, net->grads + c.w_off));
    // conv1 has no BatchNorm: its LeakyReLU backward rides in the epilogue of conv2's input-gradient GEMM when the
    // kernel supports it (same arithmetic, one 3-tensor HBM pass less)
    const bool fuse1 = (i == 2);
    // the gradient this GEMM produces enters BatchNorm + LeakyReLU of layer i - 1: its reduction rides in the GEMM's epilogue
    BwdFuse* pbf = nullptr;
    if (i >= 3) {
      pbf = &net->bwd_pending[net->dbn[i - 1].id];
      *pbf = bwd_fuse_plan(net, s, net->dbn[i - 1], net->slot(s, net->oRd[i - 1]), pix * 4, pix / 256 * 4, 0.2f);
    }
    // ... and conv1's weight gradient is formed from that GEMM's tiles while they are in LDS (IgemmArgs::c1w_*); without an input
    // gradient to compute, the gradient at conv1's output is never stored
    C1WFuse* pc1w = nullptr;
    if (fuse1 && need_wgrad && net->oC1w >= 0 && gi_opt(GI_OPT_C1W_FUSE)) {
      c1w.img = (const float*)net->slot(s, net->oX); c1w.part = (float*)net->shared(net->oC1w); c1w.part_floats = net->c1w_floats - 64 * 1024;
      c1w.scale = iLS; c1w.skip_out = dx ? 0 : 1;
      pc1w = &c1w;
    }
    GI_TRY(igemm(net, 1, D, c.ca, c.ca, 0, phase_ptr(net, c), net->shared(net->ogA[i - 1]), c.cb, c.cb, 0, n, Hs, Ws, 0, GI_ACT_NONE, false,
                 nullptr, 0, fuse1 ? net->slot(s, net->oA[1]) : nullptr, 64, 0.2f, fuse1 ? &lrelu1_done : nullptr, nullptr, 0, nullptr, nullptr, pbf, nullptr,
                 nullptr, (fuse1 && net->slot_bits1[s] && gi_opt(GI_OPT_MASK_BITS)) ? (const unsigned long long*)net->slot(s, net->oBits1) : nullptr, pc1w));
  }
  if (phase == 1) return GI_OK;
  const int64_t pix = (int64_t)n * (H / 2) * (W / 2);
  void* D1 = D = side_dz(net);
  if (lrelu1_done) {
    D1 = net->shared(net->ogA[1]);
  } else {
    GI_TRY(act_bn_bwd(net, s, net->shared(net->ogA[1]), 64, 0, nullptr, 0, 0, net->slot(s, net->oA[1]), 64, 0, nullptr, D, pix, 64, GI_ACT_LRELU,
                      1.f, nullptr, need_wgrad));
  }
  if (need_wgrad && c1w.applied)
    GI_TRY(op_c1_wgrad_reduce(st, c1w.part, net->grads + net->dconv[1].w_off, 64 * 16, c1w.blocks, c1w.part + (int64_t)c1w.blocks * 1024,
                              net->c1w_floats - (int64_t)c1w.blocks * 1024));
  else if (need_wgrad)
    GI_TRY(op_c1_wgrad(st, dt, D1, (const float*)net->slot(s, net->oX), net->grads + net->dconv[1].w_off, n, H / 2, W / 2, 64, 64, 0, 0, iLS, 1.f,
                       nullptr, (float*)net->shared(net->oPart), net->part_floats));
  if (dx) GI_TRY(op_c1_scatter(st, dt, D1, net->params + net->dconv[1].w_off, nullptr, dx, n, H / 2, W / 2, 64, 64, 0, 0, 0, iLS,
                               net->shared(net->oCol)));
  return GI_OK;
}

}  // namespace

namespace {

// WGAN-GP extension (NOT in the reference): accumulates d/dtheta of
//   lam * mean_n ( || grad_x sum_m D(x)_m ||_2 - 1 )^2
// into the bound gradients. Formulation: g = grad_x sum D (one frozen backward); v = d(penalty)/dg;
// d(penalty)/dtheta = d/dtheta [ v . g ] = d/dtheta [ JVP of sum D along v ]. The JVP (tangent forward)
// and the reverse pass over the tangent graph reuse the convolution kernels; BatchNorm in train mode
// couples the tangent to the PRIMAL activations through the batch statistics, which injects a primal
// gradient at every norm layer (op_bn_tangent_inject). Tangent-side and primal-side gradients are
// stacked as a 2n batch so every layer needs one input-gradient GEMM and one weight-gradient GEMM.
int patchgan_gradient_penalty(gi_net* net, const float* xhat, int n, float lam, float* penalty_out) {
  hipStream_t st = net->ctx->stream;
  const int dt = net->dtype, H = net->H, W = net->W, s = net->gp_slot;
  GI_REQUIRE(dt == GI_F32, "gradient_penalty: built for fp32 critics (BASELINE config 2); fp16 needs tangent scaling");
  GI_REQUIRE(!net->sigmoid && net->train, "gradient_penalty: needs a train-mode critic without sigmoid");
  GI_REQUIRE(net->bn_groups == 1, "internal: gradient_penalty runs with one BatchNorm group");
  const int64_t T = 4;
  const int chans[5] = {1, 64, 128, 256, 512};
  auto A2 = [&](int i, int half) { return (void*)(net->shared(net->oA2[i]) + (int64_t)half * n * (H >> i) * (W >> i) * chans[i] * T); };
  auto G2 = [&](int i, int half) { return (void*)(net->shared(net->oG2[i]) + (int64_t)half * n * (H >> i) * (W >> i) * chans[i] * T); };
  float* gimg = (float*)net->shared(net->oGimg);
  float* vimg = (float*)net->shared(net->oVimg);
  float* ones = (float*)net->shared(net->oGPs);
  float* sumsq = ones + (n + 16);
  float* th = (float*)net->shared(net->oTh);
  float* dth = th + (int64_t)n * net->P;
  float* ytmp = sumsq;   // (n) critic outputs, overwritten by sumsq afterwards
  // 1) primal forward + frozen backward: g = d(sum D)/dx
  GI_TRY(patchgan_forward(net, s, xhat, ytmp, n));
  hipLaunchKernelGGL(fill_f32_kernel, dim3(1), dim3(256), 0, st, ones, n, 1.0f);
  GI_LAUNCH_CHECK();
  GI_TRY(patchgan_backward(net, s, ones, gimg, 0));
  // 2) penalty value and v = d(penalty)/dg
  GI_TRY(op_gp_direction(st, gimg, n, (int64_t)H * W, lam, sumsq, vimg, penalty_out));
  // 3) tangent forward
  void* TZ = net->shared(net->oTZ);
  {
    const int64_t cnt = (int64_t)n * (H / 2) * (W / 2) * 64;
    GI_TRY(op_c1_gather(st, dt, vimg, net->params + net->dconv[1].w_off, net->shared(net->oTX[1]), n, H / 2, W / 2, 64, 64, 0, GI_ACT_NONE, 1.f));
    GI_TRY(op_mul_slope(st, dt, net->shared(net->oTX[1]), net->slot(s, net->oA[1]), A2(1, 0), cnt));
    GI_HIP(hipMemcpyAsync(A2(1, 1), net->slot(s, net->oA[1]), cnt * T, hipMemcpyDeviceToDevice, st));
  }
  for (int i = 2; i <= 4; ++i) {
    const Conv& c = net->dconv[i];
    const int Hs = H >> i, Ws = W >> i;
    const int64_t pix = (int64_t)n * Hs * Ws, cnt = pix * c.ca;
    void* TX = net->shared(net->oTX[i]);
    GI_TRY(igemm(net, 0, A2(i - 1, 0), c.cb, c.cb, 0, packed_ptr(net, c), TX, c.ca, c.ca, 0, n, Hs, Ws, 0, GI_ACT_NONE, false, nullptr));
    GI_TRY(act_bn_bwd(net, s, TX, c.ca, 0, nullptr, 0, 0, TX, c.ca, 0, net->slot(s, net->oRd[i]), TZ, pix, c.ca, GI_ACT_NONE, 1.f,
                      &net->dbn[i], 0));   // BatchNorm Jacobian applied to the tangent
    GI_TRY(op_mul_slope(st, dt, TZ, net->slot(s, net->oA[i]), A2(i, 0), cnt));
    GI_HIP(hipMemcpyAsync(A2(i, 1), net->slot(s, net->oA[i]), cnt * T, hipMemcpyDeviceToDevice, st));
  }
  // head: phi = sum_n wl . (w5 * ta4)
  HeadArgs h;
  h.a4 = A2(4, 0); h.w5 = net->params + net->w5_off; h.wl = net->params + net->wl_off; h.bl = net->params + net->bl_off;
  h.h = th; h.out = ytmp; h.n = n; h.Hh = net->Hh; h.Wh = net->Wh; h.c = 512; h.sigmoid = 0;
  GI_TRY(op_head_forward(st, dt, h));
  // 4) reverse pass over the tangent graph (seed d(phi)/d(tz) = 1; the Linear bias does not enter phi)
  HeadBwdArgs hb;
  hb.a4 = A2(4, 0); hb.w5 = h.w5; hb.wl = h.wl; hb.h = th; hb.out = ytmp; hb.dy = ones; hb.da4 = G2(4, 0);
  hb.dw5 = net->grads + net->w5_off; hb.dwl = net->grads + net->wl_off; hb.dbl = nullptr; hb.dh = dth;
  hb.n = n; hb.Hh = net->Hh; hb.Wh = net->Wh; hb.c = 512; hb.sigmoid = 0; hb.loss_scale = 1.f;
  GI_TRY(op_head_backward(st, dt, hb));
  GI_HIP(hipMemsetAsync(G2(4, 1), 0, (size_t)n * net->Hh * net->Wh * 512 * T, st));   // no primal gradient enters above the head
  char* D2 = net->shared(net->oD2);
  for (int i = 4; i >= 2; --i) {
    const Conv& c = net->dconv[i];
    const int Hs = H >> i, Ws = W >> i;
    const int64_t pix = (int64_t)n * Hs * Ws, half = pix * c.ca * T;
    // tangent-gradient chain: d(tx_i) = J_BN( slope * d(ta_i) )
    GI_TRY(act_bn_bwd(net, s, G2(i, 0), c.ca, 0, nullptr, 0, 0, net->slot(s, net->oA[i]), c.ca, 0, net->slot(s, net->oRd[i]), D2, pix, c.ca,
                      GI_ACT_LRELU, 1.f, &net->dbn[i], 0));
    // primal chain: standard BatchNorm backward of the gradient arriving from the layer above
    GI_TRY(act_bn_bwd(net, s, G2(i, 1), c.ca, 0, nullptr, 0, 0, net->slot(s, net->oA[i]), c.ca, 0, net->slot(s, net->oRd[i]), D2 + half, pix,
                      c.ca, GI_ACT_LRELU, 1.f, &net->dbn[i], 1));
    // dependence of the BatchNorm Jacobian on the primal input
    BNPtrs bp = bn_ptrs(net, s, net->dbn[i]);
    GI_TRY(op_bn_tangent_inject(st, dt, G2(i, 0), net->slot(s, net->oA[i]), net->shared(net->oTX[i]), net->slot(s, net->oRd[i]), D2 + half, pix,
                                c.ca, net->params + net->dbn[i].gamma_off, bp.mean, bp.inv, net->grads + net->dbn[i].gamma_off,
                                (float*)net->shared(net->oGPpart), (float*)net->shared(net->oGPsums)));
    // conv_i on the stacked 2n batch: [d(tx_i); dxp_i] x [ta_{i-1}; a_{i-1}]
    GI_TRY(wgrad(net, D2, c.ca, c.ca, 0, 0, A2(i - 1, 0), c.cb, c.cb, 0, 2 * n, Hs, Ws, net->grads + c.w_off));
    GI_TRY(igemm(net, 1, D2, c.ca, c.ca, 0, phase_ptr(net, c), G2(i - 1, 0), c.cb, c.cb, 0, 2 * n, Hs, Ws, 0, GI_ACT_NONE, false, nullptr));
  }
  {
    const int64_t pix = (int64_t)n * (H / 2) * (W / 2), half = pix * 64 * T;
    GI_TRY(act_bn_bwd(net, s, G2(1, 0), 64, 0, nullptr, 0, 0, net->slot(s, net->oA[1]), 64, 0, nullptr, D2, pix, 64, GI_ACT_LRELU, 1.f, nullptr, 0));
    GI_TRY(act_bn_bwd(net, s, G2(1, 1), 64, 0, nullptr, 0, 0, net->slot(s, net->oA[1]), 64, 0, nullptr, D2 + half, pix, 64, GI_ACT_LRELU, 1.f,
                      nullptr, 0));
    GI_TRY(op_c1_wgrad(st, dt, D2, vimg, net->grads + net->dconv[1].w_off, n, H / 2, W / 2, 64, 64, 0, 0, 1.f, 1.f, nullptr, (float*)net->shared(net->oPart), net->part_floats));
    GI_TRY(op_c1_wgrad(st, dt, D2 + half, (const float*)net->slot(s, net->oX), net->grads + net->dconv[1].w_off, n, H / 2, W / 2, 64, 64, 0, 0,
                       1.f, 1.f, nullptr, (float*)net->shared(net->oPart), net->part_floats));
  }
  return GI_OK;
}

}  // namespace

extern "C" int gi_patchgan_gradient_penalty(gi_net* net, const float* xhat, int n, float lam, float* penalty_out) {
  GI_REQUIRE(net && net->bound && net->kind == 1, "gradient_penalty: bound discriminator handle required");
  GI_REQUIRE(xhat && n >= 1 && n <= net->max_n, "gradient_penalty: n=%d (max %d)", n, net->max_n);
  // the interpolates are ONE BatchNorm population whatever the stacked real|fake passes around this call use
  // (gi_net_set_bn_groups): the penalty's primal forward, its backward and the tangent passes all run with one group
  const int groups = net->bn_groups;
  net->bn_groups = 1;
  const int rc = patchgan_gradient_penalty(net, xhat, n, lam, penalty_out);
  net->bn_groups = groups;
  return rc;
}

// flat fp32 gradient buffer of a bound handle (comm.hip)
float* gi_net_grads_ptr(gi_net* net, int64_t* floats) {
  if (!net || !net->bound) return nullptr;
  if (floats) *floats = net->n_params;
  return net->grads;
}

extern "C" int gi_net_forward(gi_net* net, int slot, const float* x, float* y, int n) {
  GI_REQUIRE(net && net->bound, "net_forward: net not bound");
  GI_REQUIRE(x && y && n >= 1 && n <= net->max_n, "net_forward: n=%d (max %d)", n, net->max_n);
  GI_REQUIRE(slot >= 0 && slot < net->n_slots, "net_forward: slot=%d", slot);
  return net->kind == 0 ? unet_forward(net, slot, x, y, n) : patchgan_forward(net, slot, x, y, n);
}

namespace {
// saved activation (pixels x channels of an NHWC tensor, ld / coff) -> fp32 (n,c,hw). scale != null: src is the RAW
// convolution output of a layer whose normalisation + activation a consumer applies on the fly (C1Affine / HeadArgs::scale4):
// the same fma(x, scale, shift) and activation are applied here (image i of population i / n_per_group, vectors gstride apart)
template <typename T>
__global__ void __launch_bounds__(256) export_act_kernel(const T* __restrict__ src, int ld, int coff, float* __restrict__ out, int n, int c, int hw,
                                                         const float* __restrict__ scale, const float* __restrict__ shift, int act,
                                                         int n_per_group, int gstride) {
  const int64_t total = (int64_t)n * c * hw;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int p = (int)(i % hw);
    const int64_t t = i / hw;
    const int ch = (int)(t % c);
    const int64_t img = t / c;
    float v = (float)src[(img * hw + p) * ld + coff + ch];
    if (scale) {
      const int g = n_per_group > 0 ? (int)(img / n_per_group) : 0;
      v = fmaf(v, scale[(int64_t)g * gstride + ch], shift[(int64_t)g * gstride + ch]);
      v = act == GI_ACT_RELU ? fmaxf(v, 0.f) : (act == GI_ACT_LRELU ? (v > 0.f ? v : 0.2f * v) : v);
      v = (float)(T)v;
    }
    out[i] = v;
  }
}
}  // namespace

// Debug: the split-K tile tickets of this network (igemm7 / igemm fix-up hand-off) that are NOT zero. Every launch leaves them zero (the
// last arriver of a tile resets its ticket); a non-zero ticket between launches means a launch was cut short and the next split-K
// layer would mis-count its arrivals. Synchronises the context's stream.
extern "C" int gi_net_debug_nonzero_tickets(gi_net* net, int* count) {
  GI_REQUIRE(net && net->bound && count, "debug_nonzero_tickets: net not bound / null output");
  *count = 0;
  if (net->oTickets < 0) return GI_OK;
  std::vector<unsigned> h(GI_IGEMM_TICKETS);
  GI_HIP(hipMemcpyAsync(h.data(), net->shared(net->oTickets), (size_t)GI_IGEMM_TICKETS * 4, hipMemcpyDeviceToHost, net->ctx->stream));
  GI_HIP(hipStreamSynchronize(net->ctx->stream));
  for (unsigned v : h) *count += v != 0;
  return GI_OK;
}

extern "C" int gi_net_saved_activation(gi_net* net, int slot, int kind, int level, float* out_nchw, int64_t count) {
  GI_REQUIRE(net && net->bound && out_nchw, "saved_activation: net not bound / null output");
  GI_REQUIRE(slot >= 0 && slot < net->n_slots && net->slot_n[slot] > 0, "saved_activation: slot %d holds no forward", slot);
  GI_REQUIRE(!(net->kind == 0 && net->slot_inference[slot]), "saved_activation: slot %d holds an inference forward (nothing saved)", slot);
  const int n = net->slot_n[slot];
  const void* src = nullptr;
  int ld = 0, coff = 0, c = 0, hw = 0, act = GI_ACT_NONE, npg = 0, gstride = 0;
  const float* scale = nullptr; const float* shift = nullptr;
  if (net->kind == 0) {
    const int nd = net->nd;
    GI_REQUIRE((kind == 0 && level >= 1 && level <= nd) || (kind == 1 && level >= 1 && level <= nd - 1),
               "saved_activation: generator kind=%d level=%d (kind 0: 1..%d, kind 1: 1..%d)", kind, level, nd, nd - 1);
    c = net->ch[level]; hw = net->Hk[level] * net->Wk[level];
    if (kind == 0 && level == nd) { src = net->slot(slot, net->oE); ld = c; }
    else {
      src = net->slot(slot, net->oC[level]); ld = 2 * c; coff = kind ? c : 0;
      if (kind == 1 && level == 1 && net->slot_fused_u2[slot]) {   // never materialised: the head applies it on the fly
        BNPtrs p = bn_ptrs(net, slot, net->unorm[2]);
        src = net->slot(slot, net->oU[2]); ld = c; coff = 0; scale = p.scale; shift = p.shift; act = GI_ACT_RELU;
      }
    }
  } else {
    GI_REQUIRE(kind == 0 && level >= 1 && level <= 4, "saved_activation: discriminator kind=%d level=%d (kind 0, 1..4)", kind, level);
    const int chans[5] = {1, 64, 128, 256, 512};
    c = chans[level]; hw = (net->H >> level) * (net->W >> level); ld = c;
    src = net->slot(slot, net->oA[level]);
    if (level == 4 && net->slot_fused_u2[slot]) {
      BNPtrs p = bn_ptrs(net, slot, net->dbn[4]);
      src = net->slot(slot, net->oRd[4]); scale = p.scale; shift = p.shift; act = GI_ACT_LRELU;
      npg = n / net->slot_groups[slot]; gstride = 4 * 512;
    }
  }
  GI_REQUIRE(count == (int64_t)n * c * hw, "saved_activation: count %lld != n*c*h*w = %lld", (long long)count, (long long)n * c * hw);
  hipStream_t st = net->ctx->stream;
  if (net->dtype == GI_F16)
    hipLaunchKernelGGL(export_act_kernel<half_t>, dim3(grid1d(count)), dim3(256), 0, st, (const half_t*)src, ld, coff, out_nchw, n, c, hw, scale, shift,
                       act, npg, gstride);
  else
    hipLaunchKernelGGL(export_act_kernel<float>, dim3(grid1d(count)), dim3(256), 0, st, (const float*)src, ld, coff, out_nchw, n, c, hw, scale, shift,
                       act, npg, gstride);
  GI_LAUNCH_CHECK();
  return GI_OK;
}

// weight gradients on the second stream (side_begin), joined before the caller sees the gradients
static int backward_joined(gi_net* net, int slot, const float* dy, float* dx, int need_wgrad, int phase) {
  GI_TRY(side_begin(net, need_wgrad));
  const int rc = net->kind == 0 ? unet_backward(net, slot, dy, dx, need_wgrad, phase) : patchgan_backward(net, slot, dy, dx, need_wgrad, phase);
  const int rj = side_join(net);
  return rc != GI_OK ? rc : rj;
}

extern "C" int gi_net_backward(gi_net* net, int slot, const float* dy, float* dx, int need_wgrad) {
  GI_REQUIRE(net && net->bound, "net_backward: net not bound");
  GI_REQUIRE(dy, "net_backward: dy is null");
  GI_REQUIRE(slot >= 0 && slot < net->n_slots, "net_backward: slot=%d", slot);
  return backward_joined(net, slot, dy, dx, need_wgrad, 0);
}

extern "C" int gi_net_backward_phase(gi_net* net, int slot, const float* dy, float* dx, int need_wgrad, int phase) {
  GI_REQUIRE(net && net->bound, "net_backward_phase: net not bound");
  GI_REQUIRE(dy, "net_backward_phase: dy is null");
  GI_REQUIRE(slot >= 0 && slot < net->n_slots, "net_backward_phase: slot=%d", slot);
  GI_REQUIRE(phase >= 0 && phase <= 4, "net_backward_phase: phase=%d", phase);
  if (net->kind != 0 && phase >= 3) return GI_OK;   // discriminator: two phases
  return backward_joined(net, slot, dy, dx, need_wgrad, phase);
}

// first float of the gradient region completed by phase 1 (generator: the innermost up-conv weight; discriminator:
// the conv4 weight)
extern "C" int64_t gi_net_phase_split(gi_net* net) {
  if (!net) return GI_ERR_INVALID;
  if (net->kind != 0) return net->dconv[4].w_off;
  return net->up[net->nd].w_off;
}
// generator: first float of the region completed by phase 3 (the level-5 down-conv weight); phases 3 + 4 == phase 2
extern "C" int64_t gi_net_phase_split2(gi_net* net) {
  if (!net) return GI_ERR_INVALID;
  if (net->kind != 0) return 0;
  return net->conv[net->nd < 5 ? net->nd : 5].w_off;
}
