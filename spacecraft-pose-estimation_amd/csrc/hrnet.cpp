// Host-side plan of the pose_hrnet forward: builds the launch list from the model description,
// folds BatchNorm into the convolutions, packs weights for the MFMA kernels, plans the
// activation arena, and replays the launch list on a stream.
//
// Structure follows the reference landmark_regression/lib/models/pose_hrnet.py:
//   stem :282-288/:426-431, layer1 (4 Bottlenecks) :289/:374-391/:78-98,
//   transitions :333-372 (new branch always from the LAST previous branch, :445/:453),
//   HighResolutionModule branches :139-185 + fuse :187-242/:247-265,
//   stages :393-423 (last stage-4 module fuses to branch 0 only), final_layer :323-329/:458.
// desc.head != 0 selects the hrnet_cms family (lib/models/hrnet_cms.py, hrnet_cms_384.py): same trunk with
//   multi_scale_output=True in the last module (hrnet_cms.py:321-322) and four transposed-conv heads summed
//   coarse-to-fine (:353-419, :551-557) -- see head.hip.
// Checkpoint keys are the reference module's state_dict keys.
#include <math.h>
#include <stdio.h>
#include <stdarg.h>
#include <new>

#include <map>
#include <string>
#include <vector>

#include "common.h"

namespace scpose {

static const double kBnEps = 1e-5;

struct HostTensor { const float* p; int64_t n; };

struct Weights {
  std::map<std::string, HostTensor> m;
  bool allow_missing;
  std::string missing;  // first missing key

  const float* get(const std::string& k, int64_t numel) {
    auto it = m.find(k);
    if (it == m.end() || it->second.n != numel) {
      if (missing.empty()) missing = k + (it == m.end() ? "" : " (wrong size)");
      return nullptr;
    }
    return it->second.p;
  }
};

// conv(+bn) -> folded f32 weight [cout][cin][k][k] and bias [cout]
static bool fold(Weights& W, const std::string& conv, const std::string& bn, int cout, int cin,
                 int ks, bool conv_bias, std::vector<float>* w, std::vector<float>* b) {
  const int64_t per = (int64_t)cin * ks * ks;
  w->assign((size_t)cout * per, 0.f);
  b->assign(cout, 0.f);
  const float* cw = W.get(conv + ".weight", cout * per);
  const float* cb = conv_bias ? W.get(conv + ".bias", cout) : nullptr;
  const float *g = nullptr, *be = nullptr, *mu = nullptr, *var = nullptr;
  if (!bn.empty()) {
    g = W.get(bn + ".weight", cout); be = W.get(bn + ".bias", cout);
    mu = W.get(bn + ".running_mean", cout); var = W.get(bn + ".running_var", cout);
  }
  if (!W.missing.empty() && !W.allow_missing) return false;
  for (int o = 0; o < cout; ++o) {
    double s = 1.0, sh = 0.0;
    if (!bn.empty()) {
      const double gg = g ? g[o] : 1.0, bb = be ? be[o] : 0.0, mm = mu ? mu[o] : 0.0, vv = var ? var[o] : 1.0;
      s = gg / sqrt(vv + kBnEps);
      sh = bb - mm * s;
    }
    if (cw)
      for (int64_t i = 0; i < per; ++i) (*w)[o * per + i] = (float)((double)cw[o * per + i] * s);
    (*b)[o] = (float)((cb ? (double)cb[o] * s : 0.0) + sh);
  }
  return true;
}

enum OpKind { OP_STEM = 0, OP_CONV = 1, OP_FUSE = 2, OP_BLOCK = 3, OP_HEAD = 4, OP_STEM2 = 5, OP_BNECK = 6, OP_FDOWN = 8 };   // (9 is the profile signature of a branch chain: the OP_CONVs of a branch run as one launch, see scpose_hrnet::Chain)   // OP_FDOWN: fuse row 0 + first down hops of branch 0 (fuse_down.hip; 7 is the fused tail's profile signature)   // OP_STEM2: fused stem (stem_fused.hip); OP_BNECK: fused Bottleneck (bottleneck.hip)   // OP_BLOCK: fused BasicBlock (conv_block_kernel.h); OP_HEAD: head.hip

struct TensorDesc {
  int C, ds;       // channels, log2 spatial downscale w.r.t. the network input
  int last_use;    // index of the last op reading it
};

struct Op {
  int kind;
  int in, out, res;  // tensor ids (-1 none; in == -2: network input; out == -2: heatmaps)
  int in2;           // second input of a K-concatenated 1x1 conv (-1 none); stored +1 so that Op{} means none
  int conv;          // index into convs
  int conv2;         // OP_BLOCK: second convolution of the block
  int relu, out_f32;
  int nterms, term[4], shift[4];
  int head;          // OP_HEAD: index into head_bias; in = tap map, res = coarser level (f32) or -1
  int nouts, outs[3];  // OP_FDOWN: the down paths' first-hop outputs (row 1's result, rows 2 / 3's intermediates); out = fuse row 0's sum,
                     // term[] / shift[] = its low-resolution terms, conv = index into fdowns
  // Concurrency classes for the captured (hipGraph) forward: ops of one epoch on different lanes are independent
  // (the branches of a HighResolutionModule, pose_hrnet.py:247-253; the fuse rows :254-265; the transition convs
  // :333-372); epochs are separated by a join of all lanes.  The serial forward ignores both.
  int epoch, lane;
  int par_kind;      // what the epoch's lanes are: 0 serial, 1 the branches of a module, 2 fuse rows, 3 transition convolutions
};

}  // namespace scpose

// Streams and events of a forward whose independent ops run side by side (only used while capturing a hipGraph).
struct scpose_hrnet_lanes {
  hipStream_t side[3] = {nullptr, nullptr, nullptr};   // lanes 1..3 (lane 0 is the stream the forward is launched on)
  std::vector<hipEvent_t> fork;                         // one per parallel epoch, recorded on lane 0
  std::vector<hipEvent_t> join;                         // three per parallel epoch, recorded on lanes 1..3
  unsigned kinds = 0xe;                                 // bit k: epochs of Op::par_kind k run on lanes (1 branches, 2 fuse rows, 3 transitions)
};

struct scpose_hrnet {
  scpose_hrnet_desc desc;
  int device = 0;
  std::vector<scpose::PackedConv> convs;
  float* d_stem_w = nullptr;   // folded, [8][27][8] (channel group, tap, channel in group)
  float* d_stem_b = nullptr;   // [64]
  float* d_mean_std = nullptr; // [6]
  void* d_stemf_w1 = nullptr;  // fused stem (stem_fused.hip): packed conv1 / conv2 weights and biases
  void* d_stemf_w2 = nullptr;
  float* d_stemf_b1 = nullptr;
  float* d_stemf_b2 = nullptr;
  struct Bneck { void* w1 = nullptr; void* w2 = nullptr; void* w3 = nullptr; float* bias = nullptr; int cin = 256; };   // fused Bottlenecks (bottleneck.hip)
  std::vector<Bneck> bnecks;
  // branch chains (conv_chain.hip): ops[first_op .. first_op + nops) are the 3x3 convolutions of the BasicBlocks of one branch of a module;
  // at map sizes the chain kernel supports they run as ONE launch at first_op and the others launch nothing (decided per forward)
  struct Chain { int first_op = -1, nops = 0, C = 0; void* d_w = nullptr; float* d_b = nullptr; };
  std::vector<Chain> chains;
  std::vector<int> chain_at;    // per op: index into chains of the chain that STARTS there, else -1
  std::vector<scpose::FuseDownPacked> fdowns;   // fuse row 0 + first down hops of branch 0 (fuse_down.hip), one per module that qualifies
  float* d_head_bias = nullptr; // [4][16] folded biases of the hrnet_cms heads
  uint32_t* d_sched = nullptr;  // 16 zero-initialised words per op: dynamic tile queues of the persistent kernels (conv_device.h: tile_claim)
  int head_k = 0, head_s = 1;   // transposed-conv kernel / stride of the heads (heat-map = S * branch-0 size)
  // fused tail (head_fused.hip): the last fuse sum and final_layer run as one kernel -- ops[fuse_op] is skipped and
  // ops[conv_op] launches head_fused with the fuse op's terms -- unless the forward stops at the fuse op's output
  // (scpose_hrnet_forward_tap) or the shape is not supported
  struct HeadFused { bool ok = false; int fuse_op = -1, conv_op = -1; void* d_w = nullptr; float* d_b = nullptr;
                     bool active = false; } headf;
  std::vector<scpose::TensorDesc> tensors;
  std::vector<scpose::Op> ops;
  std::vector<std::pair<std::string, int>> taps;   // named intermediate tensors (scpose_hrnet_forward_tap): name -> tensor id
  // cached arena plans: [0] serial forward (a tensor is released right after its last reader), [1] captured forward
  // (releases deferred to the end of the epoch, so that ops running side by side never share memory)
  struct Plan { int n = -1, h = -1, w = -1; size_t bytes = 0, part_off = 0; std::vector<size_t> off; } plan[2];
  // per-op HIP events of the last profiled forward (ops.size()+1, created by scpose_hrnet_create)
  std::vector<hipEvent_t> events;
  bool events_valid = false;
};

namespace scpose {

struct Builder {
  scpose_hrnet* net;
  Weights* W;
  int32_t status = SCPOSE_OK;
  int epoch = 0, lane = 0;        // stamped on every op pushed (see Op::epoch)
  bool serial = true;             // true: every op opens its own epoch (stem, layer1, heads)
  int par_kind = 0;
  void push(Op op) {
    if (serial) ++epoch;
    op.epoch = epoch; op.lane = serial ? 0 : lane; op.par_kind = serial ? 0 : par_kind;
    net->ops.push_back(op);
  }
  void parallel_begin(int kind) { ++epoch; serial = false; lane = 0; par_kind = kind; }   // the ops pushed until parallel_end() share one epoch
  void parallel_end() { serial = true; lane = 0; }

  int new_tensor(int C, int ds) {
    net->tensors.push_back(TensorDesc{C, ds, -1});
    return (int)net->tensors.size() - 1;
  }
  // y = [relu](conv_bn(x) [+ res]);  returns output tensor id (or -2 for heatmaps)
  int conv(int x, const std::string& cname, const std::string& bname, int cout, int ks, int stride,
           bool relu, int res = -1, bool conv_bias = false, bool to_heatmaps = false) {
    if (status != SCPOSE_OK) return -1;
    const int cin = net->tensors[x].C;
    std::vector<float> w, b;
    if (!fold(*W, cname, bname, cout, cin, ks, conv_bias, &w, &b)) { status = SCPOSE_E_MISSING; return -1; }
    PackedConv pc;
    const int32_t st = conv_upload(w.data(), b.data(), cout, cin, ks, stride, net->desc.dtype, &pc);
    if (st != SCPOSE_OK) { status = st; return -1; }
    net->convs.push_back(pc);
    const int ds = net->tensors[x].ds + (stride == 2 ? 1 : 0);
    Op op{};
    op.kind = OP_CONV; op.in = x; op.res = res; op.conv = (int)net->convs.size() - 1;
    op.relu = relu; op.out_f32 = to_heatmaps;
    op.out = to_heatmaps ? -2 : new_tensor(cout, ds);
    push(op);
    return op.out;
  }
  // y = relu(conv_bn_a(xa) + conv_bn_b(xb)), both 1x1: ONE convolution over the concatenated input channels
  // [xa; xb] with weights [Wa | Wb] and bias ba + bb.  Used for the first Bottleneck, whose residual is itself a
  // 1x1 conv + BN of the block input (pose_hrnet.py:78-98, :374-391): the 256-channel residual tensor is then
  // never written or read.  (The sum is formed in the fp32 accumulators, i.e. without the 16-bit rounding the
  // stored residual would get.)
  int conv_cat(int xa, const std::string& ca, const std::string& bna, int xb, const std::string& cb,
               const std::string& bnb, int cout, bool relu) {
    if (status != SCPOSE_OK) return -1;
    const int c1 = net->tensors[xa].C, c2 = net->tensors[xb].C;
    std::vector<float> w1, b1, w2, b2;
    if (!fold(*W, ca, bna, cout, c1, 1, false, &w1, &b1) || !fold(*W, cb, bnb, cout, c2, 1, false, &w2, &b2)) {
      status = SCPOSE_E_MISSING; return -1;
    }
    std::vector<float> w((size_t)cout * (c1 + c2)), b(cout);
    for (int o = 0; o < cout; ++o) {
      for (int i = 0; i < c1; ++i) w[(size_t)o * (c1 + c2) + i] = w1[(size_t)o * c1 + i];
      for (int i = 0; i < c2; ++i) w[(size_t)o * (c1 + c2) + c1 + i] = w2[(size_t)o * c2 + i];
      b[o] = b1[o] + b2[o];
    }
    PackedConv pc;
    const int32_t st = conv_upload(w.data(), b.data(), cout, c1 + c2, 1, 1, net->desc.dtype, &pc);
    if (st != SCPOSE_OK) { status = st; return -1; }
    if (pc.variant != 0 || (c1 / 8) % pc.cp != 0) { conv_free(&pc); return -2; }   // caller falls back to two convolutions
    net->convs.push_back(pc);
    Op op{};
    op.kind = OP_CONV; op.in = xa; op.in2 = xb + 1; op.res = -1; op.conv = (int)net->convs.size() - 1;
    op.relu = relu; op.out_f32 = 0;
    op.out = new_tensor(cout, net->tensors[xa].ds);
    push(op);
    return op.out;
  }
  // Bottleneck (layer1, pose_hrnet.py:78-98): one fused launch (bottleneck.hip).  Blocks 1-3 have an identity residual;
  // block 0 (64 input channels) projects its residual with `downsample` (1x1 conv + BN, :378-384), which rides in conv3.
  int bottleneck(int x, const std::string& p) {
    if (status != SCPOSE_OK) return -1;
    const int cin = net->tensors[x].C;
    const bool proj = cin == 64;
    std::vector<float> w1, b1, w2, b2, w3, b3, wd, bd;
    if (!fold(*W, p + ".conv1", p + ".bn1", 64, cin, 1, false, &w1, &b1) || !fold(*W, p + ".conv2", p + ".bn2", 64, 64, 3, false, &w2, &b2) ||
        !fold(*W, p + ".conv3", p + ".bn3", 256, 64, 1, false, &w3, &b3) ||
        (proj && !fold(*W, p + ".downsample.0", p + ".downsample.1", 256, 64, 1, false, &wd, &bd))) { status = SCPOSE_E_MISSING; return -1; }
    std::vector<uint16_t> pw1, pw2, pw3;
    std::vector<float> pb;
    bottleneck_pack(w1.data(), w2.data(), w3.data(), proj ? wd.data() : nullptr, b1.data(), b2.data(), b3.data(), proj ? bd.data() : nullptr,
                    cin, net->desc.dtype, &pw1, &pw2, &pw3, &pb);
    scpose_hrnet::Bneck bn;
    bn.cin = cin;
    auto up = [&](void** d, const void* h, size_t bytes) {
      if (hipMalloc(d, bytes) != hipSuccess || hipMemcpy(*d, h, bytes, hipMemcpyHostToDevice) != hipSuccess) status = SCPOSE_E_HIP;
    };
    up(&bn.w1, pw1.data(), pw1.size() * 2); up(&bn.w2, pw2.data(), pw2.size() * 2); up(&bn.w3, pw3.data(), pw3.size() * 2);
    up(reinterpret_cast<void**>(&bn.bias), pb.data(), pb.size() * 4);
    net->bnecks.push_back(bn);
    if (status != SCPOSE_OK) { set_error("hrnet_create: uploading the fused Bottleneck weights failed"); return -1; }
    Op op{};
    op.kind = OP_BNECK; op.in = x; op.res = -1; op.conv = (int)net->bnecks.size() - 1; op.relu = 1;
    op.out = new_tensor(256, net->tensors[x].ds);
    push(op);
    return op.out;
  }
  // Bottleneck as a STAGE block (EXTRA.STAGEk.BLOCK = BOTTLENECK, pose_hrnet.py:266-269, :142-154): 4 * planes channels in and out,
  // identity residual.  256 -> 64 -> 256 runs on the fused layer1 kernel, anything else as its three convolutions.
  int stage_bottleneck(int x, const std::string& p, int planes) {
    if (status != SCPOSE_OK) return -1;
    static const char* bn_env = dev_env("SCPOSE_BNECK_FUSED");
    if (planes == 64 && net->tensors[x].C == 256 && bottleneck_fusable(256, 64, 256) && !(bn_env && atoi(bn_env) == 0)) return bottleneck(x, p);
    int y = conv(x, p + ".conv1", p + ".bn1", planes, 1, 1, true);
    y = conv(y, p + ".conv2", p + ".bn2", planes, 3, 1, true);
    return conv(y, p + ".conv3", p + ".bn3", 4 * planes, 1, 1, true, x);
  }
  // BasicBlock relu(conv2(relu(conv1(x))) + x): one fused launch when the pair qualifies, else two convolutions
  int basic_block(int x, const std::string& p, int C) {
    if (status != SCPOSE_OK) return -1;
    const size_t first = net->convs.size(), first_op = net->ops.size();
    const int u = conv(x, p + ".conv1", p + ".bn1", C, 3, 1, true);
    const int t = conv(u, p + ".conv2", p + ".bn2", C, 3, 1, true, x);
    if (status != SCPOSE_OK || net->tensors[x].C != C) return t;
    if (!block_fusable(net->convs[first], net->convs[first + 1])) return t;
    // replace the two ops by one; the intermediate tensor u stays unused (never planned: last_use < 0)
    net->ops.resize(first_op);
    if (serial) epoch -= 2;   // the two replaced ops' epochs
    Op op{};
    op.kind = OP_BLOCK; op.in = x; op.res = -1; op.conv = (int)first; op.conv2 = (int)first + 1; op.relu = 1;
    op.out = t;
    push(op);
    return t;
  }
  // The ops pushed since first_op are the convolutions of `names.size() / 2` BasicBlocks of one branch (C -> C, 3x3, stride 1, each
  // its own OP_CONV): pack them a second time for the branch-chain kernel (conv_chain.hip).  names: (conv, bn) pairs in op order.
  void chain(size_t first_op, const std::vector<std::pair<std::string, std::string>>& names, int C) {
    if (status != SCPOSE_OK || !conv_chain_channels(C)) return;
    const size_t nops = net->ops.size() - first_op;
    if (nops != names.size() || nops < 2 || nops % 2) return;
    for (size_t k = 0; k < nops; ++k) {
      const Op& op = net->ops[first_op + k];
      if (op.kind != OP_CONV || op.in2 > 0 || op.out_f32) return;
      const PackedConv& pc = net->convs[op.conv];
      if (pc.ks != 3 || pc.stride != 1 || pc.cin != C || pc.cout != C || !op.relu) return;
      if ((k & 1) ? (op.res != net->ops[first_op + k - 1].in) : (op.res != -1)) return;     // conv2's residual is its block's input
      if (k > 0 && op.in != net->ops[first_op + k - 1].out) return;
    }
    std::vector<float> w((size_t)nops * C * C * 9), b((size_t)nops * C);
    for (size_t k = 0; k < nops; ++k) {
      std::vector<float> wk, bk;
      if (!fold(*W, names[k].first, names[k].second, C, C, 3, false, &wk, &bk)) { status = SCPOSE_E_MISSING; return; }
      memcpy(w.data() + k * C * C * 9, wk.data(), wk.size() * sizeof(float));
      memcpy(b.data() + k * C, bk.data(), (size_t)C * sizeof(float));
    }
    std::vector<uint16_t> pk(conv_chain_pack(w.data(), (int)nops, C, net->desc.dtype, nullptr) / 2);
    conv_chain_pack(w.data(), (int)nops, C, net->desc.dtype, pk.data());
    scpose_hrnet::Chain ch;
    ch.first_op = (int)first_op; ch.nops = (int)nops; ch.C = C;
    if (hipMalloc(&ch.d_w, pk.size() * 2) != hipSuccess || hipMalloc(&ch.d_b, b.size() * sizeof(float)) != hipSuccess ||
        hipMemcpy(ch.d_w, pk.data(), pk.size() * 2, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(ch.d_b, b.data(), b.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
      set_error("hrnet_create: uploading the branch-chain weights failed");
      net->chains.push_back(ch);   // (hrnet_free releases whatever was allocated)
      status = SCPOSE_E_HIP;
      return;
    }
    net->chains.push_back(ch);
  }
  // x_b = Conv2d(32->J,1x1)(ConvTranspose2d(C->32,K,S,p1,op1)(y)) [+ bilinear_x2(prev)]  (hrnet_cms.py:353-368, :551-557)
  // folded into one transposed convolution C -> J: an MFMA 1x1 convolution to the tap map + the gather of head.hip.
  // Returns the f32 tensor id of x_b (or -2 when it is the network output).
  int head(int y, const std::string& name, int b, int K, int S, int prev, bool to_heatmaps) {
    if (status != SCPOSE_OK) return -1;
    const int C = net->tensors[y].C, J = net->desc.num_joints, M = 32, cpt = J <= 8 ? 8 : 16;
    const float* wt = W->get(name + ".0.weight", (int64_t)C * M * K * K);   // ConvTranspose2d weight: (in, out, k, k)
    const float* bt = W->get(name + ".0.bias", M);
    const float* wc = W->get(name + ".1.weight", (int64_t)J * M);
    const float* bc = W->get(name + ".1.bias", J);
    if (!W->missing.empty() && !W->allow_missing) { status = SCPOSE_E_MISSING; return -1; }
    const int cout = K * K * cpt;
    std::vector<float> w((size_t)cout * C, 0.f), zero(cout, 0.f);
    if (wt && wc)
      for (int ci = 0; ci < C; ++ci)
        for (int tap = 0; tap < K * K; ++tap)
          for (int j = 0; j < J; ++j) {
            double acc = 0;
            for (int m = 0; m < M; ++m) acc += (double)wt[((size_t)ci * M + m) * K * K + tap] * wc[j * M + m];
            w[((size_t)tap * cpt + j) * C + ci] = (float)acc;
          }
    float bias[16] = {0};
    for (int j = 0; j < J; ++j) {
      double acc = bc ? bc[j] : 0.0;
      if (wc && bt) for (int m = 0; m < M; ++m) acc += (double)wc[j * M + m] * bt[m];
      bias[j] = (float)acc;
    }
    if (hipMemcpy(net->d_head_bias + b * 16, bias, sizeof(bias), hipMemcpyHostToDevice) != hipSuccess) {
      status = SCPOSE_E_HIP; return -1;
    }
    PackedConv pc;
    const int32_t st = conv_upload(w.data(), zero.data(), cout, C, 1, 1, net->desc.dtype, &pc);
    if (st != SCPOSE_OK) { status = st; return -1; }
    net->convs.push_back(pc);
    Op cv{};
    cv.kind = OP_CONV; cv.in = y; cv.res = -1; cv.conv = (int)net->convs.size() - 1;
    cv.out = new_tensor(cout, net->tensors[y].ds);
    push(cv);
    Op op{};
    op.kind = OP_HEAD; op.in = cv.out; op.res = prev; op.conv = -1; op.head = b;
    // an f32 map of J x (S*h) x (S*w) occupies as many bytes as 2*J*S*S 16-bit channels at the branch resolution
    op.out = to_heatmaps ? -2 : new_tensor(2 * J * S * S, net->tensors[y].ds);
    push(op);
    return op.out;
  }
  // Fuse row 0 and the first hop of every down path from branch 0 as one launch (fuse_down.hip; pose_hrnet.py:211-239, :256-263).
  // fp: "stageS.M.fuse_layers"; low[j - 1]: the 1x1 up-path outputs of branches j = 1..nb-1 for row 0.  Returns fuse row 0's
  // output and fills first[i - 1] with row i's first-hop tensor (row 1: its finished term; rows 2, 3: the chain's intermediate).
  int fuse_down(int x0, const std::string& fp, const std::vector<int>& low, const std::vector<int>& cur, std::vector<int>* first) {
    if (status != SCPOSE_OK) return -1;
    const int nb = (int)cur.size(), c0 = cur[0];
    std::vector<float> w((size_t)nb * c0 * c0 * 9), b((size_t)nb * c0);
    size_t wo = 0, bo = 0;
    for (int i = 1; i < nb; ++i) {
      const int cout = i == 1 ? cur[1] : c0;   // row 1's single hop ends at its width; longer chains keep branch 0's until their last hop
      std::vector<float> wi, bi;
      const std::string cn = fmt2(fp, i);
      if (!fold(*W, cn + ".0", cn + ".1", cout, c0, 3, false, &wi, &bi)) { status = SCPOSE_E_MISSING; return -1; }
      std::copy(wi.begin(), wi.end(), w.begin() + wo); wo += wi.size();
      std::copy(bi.begin(), bi.end(), b.begin() + bo); bo += bi.size();
    }
    FuseDownPacked fd;
    const int32_t st = fuse_down_upload(w.data(), b.data(), nb, c0, net->desc.dtype, &fd);
    net->fdowns.push_back(fd);   // (pushed before the status check: hrnet_free releases whatever was allocated)
    if (st != SCPOSE_OK) { status = st; return -1; }
    const int ds = net->tensors[x0].ds;
    Op op{};
    op.kind = OP_FDOWN; op.in = x0; op.res = -1; op.conv = (int)net->fdowns.size() - 1; op.relu = 1;
    op.nterms = nb - 1;
    for (int k = 0; k < nb - 1; ++k) { op.term[k] = low[k]; op.shift[k] = k + 1; }
    op.nouts = nb - 1;
    for (int i = 1; i < nb; ++i) { op.outs[i - 1] = new_tensor(i == 1 ? cur[1] : c0, ds + 1); first->push_back(op.outs[i - 1]); }
    op.out = new_tensor(c0, ds);
    push(op);
    return op.out;
  }
  static std::string fmt2(const std::string& fp, int i) {   // "<fp>.<i>.0.0": row i, source branch 0, hop 0
    char buf[256];
    snprintf(buf, sizeof(buf), "%s.%d.0.0", fp.c_str(), i);
    return buf;
  }
  int fuse(const std::vector<int>& terms, const std::vector<int>& shifts, int C, int ds) {
    if (status != SCPOSE_OK) return -1;
    Op op{};
    op.kind = OP_FUSE; op.in = -1; op.res = -1; op.conv = -1; op.relu = 1;
    op.nterms = (int)terms.size();
    for (int k = 0; k < op.nterms; ++k) { op.term[k] = terms[k]; op.shift[k] = shifts[k]; }
    op.out = new_tensor(C, ds);
    push(op);
    return op.out;
  }
};

static std::string fmt(const char* f, ...) {
  char buf[256];
  va_list ap; va_start(ap, f); vsnprintf(buf, sizeof(buf), f, ap); va_end(ap);
  return buf;
}

int32_t hrnet_build(scpose_hrnet* net, Weights& W) {
  const scpose_hrnet_desc& d = net->desc;
  Builder B{net, &W};

  // ---- stem conv1 (f32 VALU kernel, own weight format) ----
  {
    std::vector<float> w, b;
    if (!fold(W, "conv1", "bn1", 64, 3, 3, false, &w, &b)) return SCPOSE_E_MISSING;
    SCP_CHECK_HIP(hipMalloc(&net->d_stem_w, w.size() * 4));
    SCP_CHECK_HIP(hipMalloc(&net->d_stem_b, b.size() * 4));
    SCP_CHECK_HIP(hipMalloc(&net->d_mean_std, 6 * 4));
    std::vector<float> wp(w.size());   // [64][27] -> [8 groups][27][8]: the stem kernel advances channel pairs with packed FMAs
    for (int o = 0; o < 64; ++o)
      for (int k = 0; k < 27; ++k) wp[((size_t)(o / 8) * 27 + k) * 8 + o % 8] = w[(size_t)o * 27 + k];
    SCP_CHECK_HIP(hipMemcpy(net->d_stem_w, wp.data(), wp.size() * 4, hipMemcpyHostToDevice));
    SCP_CHECK_HIP(hipMemcpy(net->d_stem_b, b.data(), b.size() * 4, hipMemcpyHostToDevice));
    float ms[6] = {d.mean[0], d.mean[1], d.mean[2], d.std[0], d.std[1], d.std[2]};
    SCP_CHECK_HIP(hipMemcpy(net->d_mean_std, ms, sizeof(ms), hipMemcpyHostToDevice));
  }
  int x;
  static const char* stemf_env = dev_env("SCPOSE_STEM_FUSED");
  if (!(stemf_env && atoi(stemf_env) == 0)) {
    // conv1 + conv2 as one launch: the 64 x H/2 x W/2 tensor between them never reaches HBM (stem_fused.hip)
    std::vector<float> w1, b1, w2, b2;
    if (!fold(W, "conv1", "bn1", 64, 3, 3, false, &w1, &b1) || !fold(W, "conv2", "bn2", 64, 64, 3, false, &w2, &b2)) return SCPOSE_E_MISSING;
    std::vector<uint16_t> pw1, pw2;
    std::vector<float> pb1, pb2;
    stem_fused_pack(w1.data(), b1.data(), w2.data(), b2.data(), d.dtype, &pw1, &pw2, &pb1, &pb2);
    SCP_CHECK_HIP(hipMalloc(&net->d_stemf_w1, pw1.size() * 2));
    SCP_CHECK_HIP(hipMalloc(&net->d_stemf_w2, pw2.size() * 2));
    SCP_CHECK_HIP(hipMalloc(&net->d_stemf_b1, 64 * 4));
    SCP_CHECK_HIP(hipMalloc(&net->d_stemf_b2, 64 * 4));
    SCP_CHECK_HIP(hipMemcpy(net->d_stemf_w1, pw1.data(), pw1.size() * 2, hipMemcpyHostToDevice));
    SCP_CHECK_HIP(hipMemcpy(net->d_stemf_w2, pw2.data(), pw2.size() * 2, hipMemcpyHostToDevice));
    SCP_CHECK_HIP(hipMemcpy(net->d_stemf_b1, pb1.data(), 64 * 4, hipMemcpyHostToDevice));
    SCP_CHECK_HIP(hipMemcpy(net->d_stemf_b2, pb2.data(), 64 * 4, hipMemcpyHostToDevice));
    Op stem{};
    stem.kind = OP_STEM2; stem.in = -2; stem.res = -1; stem.conv = -1; stem.relu = 1;
    stem.out = B.new_tensor(64, 2);
    B.push(stem);
    x = stem.out;
  } else {
    Op stem{};
    stem.kind = OP_STEM; stem.in = -2; stem.res = -1; stem.conv = -1; stem.relu = 1;
    stem.out = B.new_tensor(64, 1);
    B.push(stem);
    x = stem.out;
    net->taps.emplace_back("stem1", x);
    x = B.conv(x, "conv2", "bn2", 64, 3, 2, true);
  }
  net->taps.emplace_back("stem2", x);

  // ---- layer1: 4 Bottlenecks (64 -> 256) ----
  for (int b = 0; b < 4; ++b) {
    const std::string p = fmt("layer1.%d", b);
    static const char* cat_env = dev_env("SCPOSE_CAT_DOWNSAMPLE");
    const bool cat = b == 0 && !(cat_env && atoi(cat_env) == 0);
    static const char* bn_env = dev_env("SCPOSE_BNECK_FUSED");
    static const char* bn0_env = dev_env("SCPOSE_BNECK0_FUSED");
    if (((b > 0 && net->tensors[x].C == 256) || (b == 0 && net->tensors[x].C == 64 && cat && !(bn0_env && atoi(bn0_env) == 0))) &&
        !(bn_env && atoi(bn_env) == 0)) {   // one launch per Bottleneck
      x = B.bottleneck(x, p);
      continue;
    }
    int res = x;
    if (b == 0 && !cat) res = B.conv(x, p + ".downsample.0", p + ".downsample.1", 256, 1, 1, false);
    int y = B.conv(x, p + ".conv1", p + ".bn1", 64, 1, 1, true);
    y = B.conv(y, p + ".conv2", p + ".bn2", 64, 3, 1, true);
    int nx = -2;
    if (cat) nx = B.conv_cat(y, p + ".conv3", p + ".bn3", x, p + ".downsample.0", p + ".downsample.1", 256, true);
    if (nx == -2) {   // not concatenated (disabled or the packing does not allow the split)
      if (b == 0 && cat) res = B.conv(x, p + ".downsample.0", p + ".downsample.1", 256, 1, 1, false);
      nx = B.conv(y, p + ".conv3", p + ".bn3", 256, 1, 1, true, res);
    }
    x = nx;
  }

  net->taps.emplace_back("layer1", x);
  std::vector<int> ylist{x};
  std::vector<int> pre{256};
  for (int si = 0; si < 3; ++si) {
    const int nb = d.num_branches[si];
    const bool bneck_stage = d.block[si] == 1;
    std::vector<int> cur(d.num_channels[si], d.num_channels[si] + nb);
    if (bneck_stage) for (int& c : cur) c *= 4;          // num_channels * block.expansion (pose_hrnet.py:393-400)
    const std::string tname = fmt("transition%d", si + 1);
    std::vector<int> xs;
    B.parallel_begin(3);   // transition convs: independent of each other (lane = branch they create)
    for (int i = 0; i < nb; ++i) {
      B.lane = i;
      if (i < (int)pre.size()) {
        if (cur[i] != pre[i])
          xs.push_back(B.conv(ylist.back(), fmt("%s.%d.0", tname.c_str(), i), fmt("%s.%d.1", tname.c_str(), i), cur[i], 3, 1, true));
        else
          xs.push_back(ylist[i]);
      } else {
        int t = ylist.back();
        for (int j = 0; j < i + 1 - (int)pre.size(); ++j) {
          const int cout = (j == i - (int)pre.size()) ? cur[i] : pre.back();
          t = B.conv(t, fmt("%s.%d.%d.0", tname.c_str(), i, j), fmt("%s.%d.%d.1", tname.c_str(), i, j), cout, 3, 2, true);
        }
        xs.push_back(t);
      }
    }
    B.parallel_end();
    for (int m = 0; m < d.num_modules[si]; ++m) {
      const bool multi = d.head != SCPOSE_HEAD_FINAL_LAYER || !(si == 2 && m == d.num_modules[si] - 1);
      const std::string mp = fmt("stage%d.%d", si + 2, m);
      B.parallel_begin(1);   // the branches of the module: one lane each
      for (int b = 0; b < nb; ++b) {
        B.lane = b;
        int t = xs[b];
        const size_t branch_first_op = net->ops.size();
        std::vector<std::pair<std::string, std::string>> branch_names;
        for (int k = 0; k < d.num_blocks[si][b]; ++k) {
          const std::string p = fmt("%s.branches.%d.%d", mp.c_str(), b, k);
          t = bneck_stage ? B.stage_bottleneck(t, p, d.num_channels[si][b]) : B.basic_block(t, p, cur[b]);
          branch_names.emplace_back(p + ".conv1", p + ".bn1"); branch_names.emplace_back(p + ".conv2", p + ".bn2");
        }
        if (!bneck_stage) B.chain(branch_first_op, branch_names, cur[b]);
        xs[b] = t;
      }
      B.parallel_end();
      std::vector<int> outs;
      static const char* fd_env = dev_env("SCPOSE_FUSE_DOWN");
      if (multi && !bneck_stage && fuse_down_supported(nb, cur[0], cur[1]) && !(fd_env && atoi(fd_env) == 0)) {
        // Branch 0 is read ONCE (fuse_down.hip): its fuse row and the first hop of every down path that starts from it are one
        // launch.  Two epochs instead of one: (A) lane 0: row 0's 1x1 up paths, then that launch; lanes 1..: everything of rows 1..
        // that does not depend on it -- their up paths and the down chains from branches 1..; (B) the remaining hops of the
        // chains from branch 0 and the sums of rows 1.., row i's on lane i - 1.
        std::vector<std::vector<int>> T(nb, std::vector<int>(nb, -1)), S(nb, std::vector<int>(nb, 0));
        std::vector<int> first;
        int y0 = -1;
        B.parallel_begin(2);
        for (int i = 0; i < nb; ++i) {
          B.lane = i;
          for (int j = 0; j < nb; ++j) {
            const std::string fp = fmt("%s.fuse_layers.%d.%d", mp.c_str(), i, j);
            if (j == i) {
              T[i][j] = xs[j];
            } else if (j > i) {
              T[i][j] = B.conv(xs[j], fp + ".0", fp + ".1", cur[i], 1, 1, false);
              S[i][j] = j - i;
            } else if (j >= 1) {
              int t = xs[j];
              for (int k = 0; k < i - j; ++k) {
                const bool last = k == i - j - 1;
                t = B.conv(t, fmt("%s.%d.0", fp.c_str(), k), fmt("%s.%d.1", fp.c_str(), k), last ? cur[i] : cur[j], 3, 2, !last);
              }
              T[i][j] = t;
            }
          }
          if (i == 0) y0 = B.fuse_down(xs[0], mp + ".fuse_layers", std::vector<int>(T[0].begin() + 1, T[0].end()), cur, &first);
        }
        B.parallel_end();
        if (B.status != SCPOSE_OK) break;
        outs.push_back(y0);
        B.parallel_begin(2);
        for (int i = 1; i < nb; ++i) {
          B.lane = i - 1;   // row 1 on lane 0 (the stream the epoch is captured on): nb = 2 then forks nothing, nb = 3 / 4 one lane fewer
          const std::string fp = fmt("%s.fuse_layers.%d.0", mp.c_str(), i);
          int t = first[i - 1];
          for (int k = 1; k < i; ++k) {
            const bool last = k == i - 1;
            t = B.conv(t, fmt("%s.%d.0", fp.c_str(), k), fmt("%s.%d.1", fp.c_str(), k), last ? cur[i] : cur[0], 3, 2, !last);
          }
          T[i][0] = t;
          if (B.status != SCPOSE_OK) break;
          outs.push_back(B.fuse(T[i], S[i], cur[i], net->tensors[xs[i]].ds));
        }
        B.parallel_end();
      } else {
      B.parallel_begin(2);   // the fuse rows: row i (its up / down paths, then its sum) on lane i
      for (int i = 0; i < (multi ? nb : 1); ++i) {
        B.lane = i;
        std::vector<int> terms, shifts;
        for (int j = 0; j < nb; ++j) {
          const std::string fp = fmt("%s.fuse_layers.%d.%d", mp.c_str(), i, j);
          if (j == i) {
            terms.push_back(xs[j]); shifts.push_back(0);
          } else if (j > i) {
            terms.push_back(B.conv(xs[j], fp + ".0", fp + ".1", cur[i], 1, 1, false));
            shifts.push_back(j - i);
          } else {
            int t = xs[j];
            for (int k = 0; k < i - j; ++k) {
              const bool last = k == i - j - 1;
              t = B.conv(t, fmt("%s.%d.0", fp.c_str(), k), fmt("%s.%d.1", fp.c_str(), k), last ? cur[i] : cur[j], 3, 2, !last);
            }
            terms.push_back(t); shifts.push_back(0);
          }
        }
        if (B.status != SCPOSE_OK) break;
        outs.push_back(B.fuse(terms, shifts, cur[i], net->tensors[xs[i]].ds));
      }
      B.parallel_end();
      }
      xs = outs;
      if (B.status != SCPOSE_OK) break;
      net->taps.emplace_back(mp + ".out0", xs[0]);
    }
    if (B.status != SCPOSE_OK) break;
    ylist = xs;
    pre = cur;
  }
  if (B.status == SCPOSE_OK && d.head == SCPOSE_HEAD_FINAL_LAYER) {
    B.conv(ylist[0], "final_layer", "", d.num_joints, d.final_conv_kernel, 1, false, -1, true, true);
    const int nops = (int)net->ops.size();
    if (B.status == SCPOSE_OK && d.final_conv_kernel == 1 && nops >= 2 && net->ops[nops - 2].kind == OP_FUSE &&
        net->ops[nops - 2].out == ylist[0] && net->ops[nops - 1].in == ylist[0] &&
        head_fused_supported(net->ops[nops - 2].nterms, net->tensors[ylist[0]].C, d.num_joints, 1, 32, 32)) {
      const int C = net->tensors[ylist[0]].C;
      const float* fw = W.get("final_layer.weight", (int64_t)d.num_joints * C);
      const float* fb = W.get("final_layer.bias", d.num_joints);
      if (fw && fb) {
        std::vector<uint16_t> wf(2 * 4 * 16 * 8);
        float bias[16];
        head_fused_pack(fw, fb, d.num_joints, C, d.dtype, wf.data(), bias);
        SCP_CHECK_HIP(hipMalloc(&net->headf.d_w, wf.size() * 2));
        SCP_CHECK_HIP(hipMalloc(&net->headf.d_b, sizeof(bias)));
        SCP_CHECK_HIP(hipMemcpy(net->headf.d_w, wf.data(), wf.size() * 2, hipMemcpyHostToDevice));
        SCP_CHECK_HIP(hipMemcpy(net->headf.d_b, bias, sizeof(bias), hipMemcpyHostToDevice));
        net->headf.ok = true; net->headf.fuse_op = nops - 2; net->headf.conv_op = nops - 1;
      }
    }
  } else if (B.status == SCPOSE_OK) {
    const bool cms = d.head == SCPOSE_HEAD_CMS;
    net->head_k = cms ? 5 : 3;
    net->head_s = cms ? 4 : 2;
    SCP_CHECK_HIP(hipMalloc(&net->d_head_bias, 4 * 16 * sizeof(float)));
    int prev = -1;
    for (int b = 3; b >= 0; --b) {      // hrnet_cms.py:551-557: x4, x3 = head3 + up(x4), x2 = ..., x = head + up(x2)
      const std::string name = fmt("final_layer%s_%s", b == 0 ? "" : fmt("%d", b + 1).c_str(), cms ? "equal_to_image" : "4x");
      prev = B.head(ylist[b], name, b, net->head_k, net->head_s, prev, b == 0);
    }
  }
  if (B.status == SCPOSE_E_MISSING || (!W.missing.empty() && !W.allow_missing)) {
    set_error("checkpoint tensor missing: %s", W.missing.c_str());
    return SCPOSE_E_MISSING;
  }
  if (B.status != SCPOSE_OK) return B.status;
  SCP_CHECK_HIP(hipMalloc(&net->d_sched, net->ops.size() * 16 * sizeof(uint32_t)));
  SCP_CHECK_HIP(hipMemset(net->d_sched, 0, net->ops.size() * 16 * sizeof(uint32_t)));
  net->chain_at.assign(net->ops.size(), -1);
  for (size_t c = 0; c < net->chains.size(); ++c) net->chain_at[net->chains[c].first_op] = (int)c;

  // liveness
  for (size_t i = 0; i < net->ops.size(); ++i) {
    const Op& op = net->ops[i];
    auto use = [&](int t) { if (t >= 0) net->tensors[t].last_use = (int)i; };
    use(net->ops[i].in2 - 1);
    use(op.in); use(op.res);
    for (int k = 0; k < op.nterms; ++k) use(op.term[k]);
  }
  return SCPOSE_OK;
}

// does a branch chain start at op oi, and does its kernel run at this input size?
static bool hrnet_chain_active(const scpose_hrnet* net, int oi, int h, int w) {
  if (oi < 0 || (size_t)oi >= net->chain_at.size() || net->chain_at[oi] < 0) return false;
  const scpose_hrnet::Chain& ch = net->chains[net->chain_at[oi]];
  const TensorDesc& ti = net->tensors[net->ops[oi].in];
  return conv_chain_supported(ch.C, h >> ti.ds, w >> ti.ds);
}
// ... or is op oi one of the later convolutions of such a chain (it launches nothing)?
static bool hrnet_chain_member(const scpose_hrnet* net, int oi, int h, int w) {
  for (const scpose_hrnet::Chain& ch : net->chains)
    if (oi > ch.first_op && oi < ch.first_op + ch.nops) return hrnet_chain_active(net, ch.first_op, h, w);
  return false;
}

static size_t tensor_bytes(const TensorDesc& t, int n, int h, int w) {
  const size_t b = (size_t)n * t.C * (h >> t.ds) * (w >> t.ds) * 2;
  return (b + 255) & ~(size_t)255;
}

// Greedy first-fit arena: an output is placed when its producer runs and freed after its last reader (mode 0, the
// serial forward) or at the end of the epoch of its last reader (mode 1, the captured forward whose lanes run side by side).
size_t hrnet_plan(scpose_hrnet* net, int n, int h, int w, int mode = 0) {
  scpose_hrnet::Plan& P = net->plan[mode];
  if (P.n == n && P.h == h && P.w == w) return P.bytes;
  struct Blk { size_t off, size; };
  std::vector<Blk> freel;
  size_t top = 0;
  auto alloc = [&](size_t sz) {
    int best = -1;
    for (size_t i = 0; i < freel.size(); ++i)
      if (freel[i].size >= sz && (best < 0 || freel[i].size < freel[best].size)) best = (int)i;
    if (best >= 0) {
      const size_t off = freel[best].off;
      if (freel[best].size == sz) freel.erase(freel.begin() + best);
      else { freel[best].off += sz; freel[best].size -= sz; }
      return off;
    }
    // grow: extend a trailing free block if there is one
    for (size_t i = 0; i < freel.size(); ++i)
      if (freel[i].off + freel[i].size == top) {
        const size_t off = freel[i].off;
        top = off + sz;
        freel.erase(freel.begin() + i);
        return off;
      }
    const size_t off = top;
    top += sz;
    return off;
  };
  auto release = [&](size_t off, size_t sz) {
    freel.push_back(Blk{off, sz});
    bool merged = true;
    while (merged) {
      merged = false;
      for (size_t i = 0; i < freel.size() && !merged; ++i)
        for (size_t j = 0; j < freel.size() && !merged; ++j)
          if (i != j && freel[i].off + freel[i].size == freel[j].off) {
            freel[i].size += freel[j].size;
            freel.erase(freel.begin() + j);
            merged = true;
          }
    }
  };
  P.off.assign(net->tensors.size(), 0);
  std::vector<char> released(net->tensors.size(), 0);
  std::vector<int> pending;   // mode 1: tensors whose last reader ran in the current epoch
  int chain_out = -1;         // output tensor of the branch chain being walked (allocated at the chain's first op)
  int cur_epoch = net->ops.empty() ? 0 : net->ops[0].epoch;
  for (size_t i = 0; i < net->ops.size(); ++i) {
    const Op& op = net->ops[i];
    if (mode == 1 && op.epoch != cur_epoch) {
      for (int t : pending) release(P.off[t], tensor_bytes(net->tensors[t], n, h, w));
      pending.clear();
      cur_epoch = op.epoch;
    }
    // A branch chain (conv_chain.hip) launched at its first op writes the LAST op's output tensor, frame by frame, while other
    // workgroups still read the chain's input: that output is born here, beside the (still live) input, not at the last op
    if (hrnet_chain_active(net, (int)i, h, w)) {
      const int t_out = net->ops[i + net->chains[net->chain_at[i]].nops - 1].out;
      P.off[t_out] = alloc(tensor_bytes(net->tensors[t_out], n, h, w));
      chain_out = t_out;
    }
    if (op.out >= 0 && op.out != chain_out) P.off[op.out] = alloc(tensor_bytes(net->tensors[op.out], n, h, w));
    for (int k = 0; k < op.nouts; ++k) P.off[op.outs[k]] = alloc(tensor_bytes(net->tensors[op.outs[k]], n, h, w));
    auto done = [&](int t) {
      if (t >= 0 && net->tensors[t].last_use == (int)i && !released[t]) {
        released[t] = 1;   // guard against double release (same tensor twice in one op)
        if (mode == 1) pending.push_back(t);
        else release(P.off[t], tensor_bytes(net->tensors[t], n, h, w));
      }
    };
    done(op.in); done(op.res); done(op.in2 - 1);
    for (int k = 0; k < op.nterms; ++k) done(op.term[k]);
    // an output nobody reads (cannot happen in a well-formed net) is simply never reused
  }
  // Behind the arena: the fused tail's partial maxima (head_fused.hip: part_v, then part_i).  They live in the CALLER's
  // workspace, like every other buffer a launch writes, so a captured graph -- which owns its workspace -- never holds a
  // pointer into memory a later forward of another batch size could free (they used to be engine-owned and
  // re-allocated on demand).
  P.part_off = top;
  if (net->headf.ok) top += 2 * head_fused_part_bytes(n);
  P.n = n; P.h = h; P.w = w; P.bytes = top;
  return top;
}

// Key points straight from the forward (scpose_hrnet_forward_decode): the decode of lib/core/inference.py:49-79 runs inside
// the fused tail when the network has one, else as decode.hip's kernel on the heat-maps behind the last layer.
struct HeadDecode {
  const float* center;
  const float* scale;
  int post_process;
  float* preds;
};

// does a forward of this shape run the fused tail (head_fused.hip)?
bool hrnet_tail_fused(const scpose_hrnet* net, int n, int h, int w) {
  if (!net->headf.ok) return false;
  static const char* e = dev_env("SCPOSE_NO_HEAD_FUSE");
  const Op& fo = net->ops[net->headf.fuse_op];
  const TensorDesc& ty = net->tensors[fo.out];
  return !(e && atoi(e)) && head_fused_supported(fo.nterms, ty.C, net->desc.num_joints, n, h >> ty.ds, w >> ty.ds);
}

int32_t hrnet_forward(scpose_hrnet* net, const void* in, int in_fmt, int n, int h, int w,
                      float* heatmaps, void* ws, size_t ws_bytes, hipStream_t st, bool profile, int stop_tensor = -1,
                      int mode = 0, const scpose_hrnet_lanes* lanes = nullptr, const HeadDecode* dec = nullptr) {
  SCP_REQUIRE(n > 0, "hrnet_forward: batch %d", n);
  const bool fused_tail = hrnet_tail_fused(net, n, h, w) && stop_tensor != net->ops[net->headf.fuse_op].out;
  net->headf.active = fused_tail;
  SCP_REQUIRE(heatmaps || (dec && fused_tail) || stop_tensor >= 0,
              "hrnet_forward: this network's tail is not fused for this shape -- key points need a heat-map buffer");
  SCP_REQUIRE(h % 32 == 0 && w % 32 == 0 && h > 0 && w > 0, "hrnet_forward: H=%d W=%d must be multiples of 32", h, w);
  const size_t need = hrnet_plan(net, n, h, w, mode);
  if (ws_bytes < need || !ws) {
    set_error("hrnet_forward: workspace %zu bytes < required %zu", ws_bytes, need);
    return SCPOSE_E_WORKSPACE;
  }
  char* base = static_cast<char*>(ws);
  const std::vector<size_t>& off = net->plan[mode].off;
  auto ptr = [&](int t) -> void* { return t >= 0 ? base + off[t] : nullptr; };
  float* const part_v = reinterpret_cast<float*>(base + net->plan[mode].part_off);
  int32_t* const part_i = reinterpret_cast<int32_t*>(base + net->plan[mode].part_off + head_fused_part_bytes(n));
  SCP_REQUIRE(!profile || net->events.size() == net->ops.size() + 1, "hrnet_forward: profiling events missing");
  net->events_valid = false;
  size_t opi = 0;
  hipStream_t const st0 = st;
  int cur_epoch = -1, par_index = -1;
  unsigned open_lanes = 0;   // side lanes forked in the current epoch
  int chain_end = 0;         // ops below this index belong to a branch chain that has been launched
  auto join_lanes = [&]() -> int32_t {
    for (int l = 1; l < 4; ++l)
      if (open_lanes & (1u << l)) {
        hipEvent_t e = lanes->join[(size_t)par_index * 3 + (l - 1)];
        SCP_CHECK_HIP(hipEventRecord(e, lanes->side[l - 1]));
        SCP_CHECK_HIP(hipStreamWaitEvent(st0, e, 0));
      }
    open_lanes = 0;
    return SCPOSE_OK;
  };
  for (size_t oi = 0; oi < net->ops.size(); ++oi) {
    const Op& op = net->ops[oi];
    if (lanes && op.epoch != cur_epoch) {   // epoch boundary: join the lanes of the previous epoch, fork those of this one
      { const int32_t rc = join_lanes(); if (rc != SCPOSE_OK) return rc; }
      cur_epoch = op.epoch;
      unsigned used = 0;
      for (size_t k = oi; k < net->ops.size() && net->ops[k].epoch == cur_epoch; ++k)
        if (lanes->kinds & (1u << net->ops[k].par_kind)) used |= 1u << net->ops[k].lane;
      used &= ~1u;
      if (used) {
        ++par_index;
        SCP_REQUIRE((size_t)par_index < lanes->fork.size(), "hrnet_forward: lane events missing");
        SCP_CHECK_HIP(hipEventRecord(lanes->fork[par_index], st0));
        for (int l = 1; l < 4; ++l)
          if (used & (1u << l)) SCP_CHECK_HIP(hipStreamWaitEvent(lanes->side[l - 1], lanes->fork[par_index], 0));
        open_lanes = used;
      }
    }
    st = (lanes && op.lane > 0 && (lanes->kinds & (1u << op.par_kind))) ? lanes->side[op.lane - 1] : st0;
    if (profile) SCP_CHECK_HIP(hipEventRecord(net->events[opi], st));
    ++opi;
    int32_t rc = SCPOSE_OK;
    if ((int)oi < chain_end) {
      // nothing: a convolution inside a branch chain that was launched at the chain's first op
      if (stop_tensor >= 0 && op.out == stop_tensor && (int)oi + 1 < chain_end) { set_error("hrnet_forward: tensor %d lies inside a branch chain", stop_tensor); return SCPOSE_E_INVALID; }
    } else if (hrnet_chain_active(net, (int)oi, h, w)) {
      const scpose_hrnet::Chain& ch = net->chains[net->chain_at[oi]];
      const TensorDesc& ti = net->tensors[op.in];
      chain_end = ch.first_op + ch.nops;
      rc = conv_chain_launch(ptr(op.in), ptr(net->ops[chain_end - 1].out), ch.d_w, ch.d_b, ch.nops, n, ch.C, h >> ti.ds, w >> ti.ds,
                             net->desc.dtype, net->d_sched + 16 * oi, st);
    } else if (fused_tail && (int)oi == net->headf.fuse_op) {
      // nothing: its sum is formed in registers by the next op
    } else if (fused_tail && (int)oi == net->headf.conv_op) {
      const Op& fo = net->ops[net->headf.fuse_op];
      const TensorDesc& ty = net->tensors[fo.out];
      const void* terms[4];
      for (int k = 0; k < fo.nterms; ++k) terms[k] = ptr(fo.term[k]);
      rc = head_fused_launch(terms, fo.shift, fo.nterms, n, ty.C, h >> ty.ds, w >> ty.ds, net->desc.num_joints, net->desc.dtype,
                             net->headf.d_w, net->headf.d_b, heatmaps, dec ? part_v : nullptr,
                             dec ? part_i : nullptr, dec ? dec->center : nullptr, dec ? dec->scale : nullptr,
                             dec ? dec->post_process : 0, dec ? dec->preds : nullptr, st);
    } else if (op.kind == OP_STEM) {
      rc = stem_launch(in, in_fmt, net->d_stem_w, net->d_stem_b, net->d_mean_std, n, h, w,
                       net->desc.dtype, ptr(op.out), st);
    } else if (op.kind == OP_STEM2) {
      rc = stem_fused_launch(in, in_fmt, net->d_stemf_w1, net->d_stemf_w2, net->d_stemf_b1, net->d_stemf_b2, net->d_mean_std,
                             n, h, w, net->desc.dtype, ptr(op.out), net->d_sched + 16 * oi, st);
    } else if (op.kind == OP_CONV) {
      const TensorDesc& ti = net->tensors[op.in];
      void* out = op.out == -2 ? static_cast<void*>(heatmaps) : ptr(op.out);
      rc = conv_launch(net->convs[op.conv], ptr(op.in), n, h >> ti.ds, w >> ti.ds, ptr(op.res),
                       op.relu, op.out_f32, out, st, op.in2 > 0 ? ptr(op.in2 - 1) : nullptr,
                       op.in2 > 0 ? net->tensors[op.in].C / 8 : 0, open_lanes ? conv_device_cus() / 2 : 0);
    } else if (op.kind == OP_HEAD) {
      const TensorDesc& ti = net->tensors[op.in];
      float* out = op.out == -2 ? heatmaps : static_cast<float*>(ptr(op.out));
      rc = head_gather_launch(ptr(op.in), net->d_head_bias + op.head * 16, static_cast<const float*>(ptr(op.res)), n,
                              net->desc.num_joints, h >> ti.ds, w >> ti.ds, net->head_k, net->head_s,
                              net->desc.dtype, out, st);
    } else if (op.kind == OP_BNECK) {
      const TensorDesc& ti = net->tensors[op.in];
      const scpose_hrnet::Bneck& bn = net->bnecks[op.conv];
      rc = bottleneck_launch(ptr(op.in), bn.w1, bn.w2, bn.w3, bn.bias, n, h >> ti.ds, w >> ti.ds, bn.cin, net->desc.dtype, ptr(op.out), net->d_sched + 16 * oi, st);
    } else if (op.kind == OP_FDOWN) {
      const TensorDesc& ti = net->tensors[op.in];
      const void* terms[3] = {nullptr, nullptr, nullptr};
      void* outs[3] = {nullptr, nullptr, nullptr};
      for (int k = 0; k < op.nterms; ++k) terms[k] = ptr(op.term[k]);
      for (int k = 0; k < op.nouts; ++k) outs[k] = ptr(op.outs[k]);
      rc = fuse_down_launch(net->fdowns[op.conv], ptr(op.in), n, h >> ti.ds, w >> ti.ds, terms, ptr(op.out), outs, st);
    } else if (op.kind == OP_BLOCK) {
      const TensorDesc& ti = net->tensors[op.in];
      rc = block_launch(net->convs[op.conv], net->convs[op.conv2], ptr(op.in), n, h >> ti.ds, w >> ti.ds, ptr(op.out), st);
    } else {
      const TensorDesc& to = net->tensors[op.out];
      const void* terms[4];
      for (int k = 0; k < op.nterms; ++k) terms[k] = ptr(op.term[k]);
      rc = fuse_sum_launch(terms, op.shift, op.nterms, n, to.C, h >> to.ds, w >> to.ds,
                           net->desc.dtype, ptr(op.out), st);
    }
    if (rc != SCPOSE_OK) return rc;
    if (stop_tensor >= 0 && op.out == stop_tensor) break;   // scpose_hrnet_forward_tap: the tensor is still intact in the arena
  }
  if (lanes) { const int32_t rc = join_lanes(); if (rc != SCPOSE_OK) return rc; }
  st = st0;
  if (dec && !fused_tail && stop_tensor < 0) {   // unfused tail: decode.hip on the heat-maps
    int32_t hh = 0, hw = 0;
    hh = h / 4 * net->head_s; hw = w / 4 * net->head_s;
    const int32_t rc = decode_launch(heatmaps, n, net->desc.num_joints, hh, hw, dec->center, dec->scale, dec->post_process,
                                     dec->preds, nullptr, nullptr, st);
    if (rc != SCPOSE_OK) return rc;
  }
  if (profile) {
    SCP_CHECK_HIP(hipEventRecord(net->events[opi], st));
    net->events_valid = true;
  }
  return SCPOSE_OK;
}

// per-op algorithmic work (per frame) and a signature identifying the kernel variant
// `unfused`: count a fused BasicBlock as SURVEY.md 8(d) counts the unfused pair (5 tensors); otherwise the bytes are the
// ones the launch itself has to move (block input once + output once), which is what a roofline of that launch needs.
static void op_work(const scpose_hrnet* net, const Op& op, int h, int w, double* f, double* by, int32_t sig[4], bool unfused = false) {
  *f = 0; *by = 0;
  sig[0] = op.kind; sig[1] = sig[2] = sig[3] = 0;
  if (!unfused && net->headf.active && &op == &net->ops[net->headf.fuse_op]) {
    // fused tail (head_fused.hip): this op launched nothing; its work is accounted under the next one
    const TensorDesc& to = net->tensors[op.out];
    sig[1] = op.nterms; sig[2] = to.C; sig[3] = to.C;
  } else if (!unfused && net->headf.active && &op == &net->ops[net->headf.conv_op]) {
    // last fuse sum + final_layer in one launch: the fuse row's terms in, the f32 heat-maps out (kind 7)
    const Op& fo = net->ops[net->headf.fuse_op];
    const TensorDesc& to = net->tensors[fo.out];
    const double ho = h >> to.ds, wo = w >> to.ds, J = net->desc.num_joints;
    *f = 2.0 * to.C * J * ho * wo;
    *by = J * ho * wo * 4;
    for (int k = 0; k < fo.nterms; ++k) *by += to.C * (ho / (1 << fo.shift[k])) * (wo / (1 << fo.shift[k])) * 2;
    sig[0] = 7; sig[1] = fo.nterms; sig[2] = to.C; sig[3] = (int)J;
  } else if (!unfused && hrnet_chain_active(net, (int)(&op - net->ops.data()), h, w)) {
    // branch chain (conv_chain.hip): every convolution of the branch in this one launch (kind 9); it reads the branch input once and
    // writes the branch output once
    const scpose_hrnet::Chain& ch = net->chains[net->chain_at[&op - net->ops.data()]];
    const TensorDesc& ti = net->tensors[op.in];
    const double hi = h >> ti.ds, wi = w >> ti.ds;
    *f = ch.nops * 2.0 * ch.C * ch.C * 9 * hi * wi;
    *by = 2.0 * ch.C * hi * wi * 2;
    sig[0] = 9; sig[1] = ch.nops; sig[2] = ch.C; sig[3] = ch.C;
  } else if (!unfused && hrnet_chain_member(net, (int)(&op - net->ops.data()), h, w)) {
    // launched nothing: accounted under the chain's first op
    sig[0] = 9; sig[1] = 0; sig[2] = net->convs[op.conv].cin; sig[3] = net->convs[op.conv].cout;
  } else if (op.kind == OP_STEM) {
    *f = 2.0 * 27 * 64 * (h / 2) * (w / 2);
    *by = (double)64 * (h / 2) * (w / 2) * 2 + 3.0 * h * w;   // u8 in (f32 in: 4x) + 16-bit out
    sig[1] = 32; sig[2] = 3; sig[3] = 64;
  } else if (op.kind == OP_STEM2) {
    // both stem convolutions; bytes: the launch reads the image and writes the H/4 x W/4 result (unfused accounting:
    // + the H/2 x W/2 tensor written once and read once)
    *f = 2.0 * 27 * 64 * (h / 2) * (w / 2) + 2.0 * 64 * 64 * 9 * (h / 4) * (w / 4);
    *by = (double)64 * (h / 4) * (w / 4) * 2 + 3.0 * h * w + (unfused ? 2.0 * 64 * (h / 2) * (w / 2) * 2 : 0.0);
    sig[1] = 32; sig[2] = 3; sig[3] = 64;   // sig[0] stays OP_STEM2 (5): the fused stem is its own kernel class
  } else if (op.kind == OP_CONV) {
    const PackedConv& pc = net->convs[op.conv];
    const TensorDesc& ti = net->tensors[op.in];
    const double hi = h >> ti.ds, wi = w >> ti.ds;
    const double ho = pc.stride == 2 ? hi / 2 : hi, wo = pc.stride == 2 ? wi / 2 : wi;
    *f = 2.0 * pc.cin * pc.cout * pc.ks * pc.ks * ho * wo;
    *by = pc.cin * hi * wi * 2 + pc.cout * ho * wo * (op.out_f32 ? 4 : 2) + (op.res >= 0 ? pc.cout * ho * wo * 2 : 0);
    if (op.in2 > 0 && unfused) *by += 2.0 * pc.cout * ho * wo * 2;   // the unfused pair writes the residual tensor once and reads it once; the K-concatenated launch does neither
    sig[1] = pc.ks * 10 + pc.stride; sig[2] = pc.cin; sig[3] = pc.cout;
  } else if (op.kind == OP_BLOCK) {
    // flops of the two convolutions; bytes: the fused launch reads x once and writes the block output once (the
    // intermediate tensor and the residual re-read of the unfused pair -- 3 of its 5 tensors -- never touch HBM)
    const PackedConv& pc = net->convs[op.conv];
    const TensorDesc& ti = net->tensors[op.in];
    const double hi = h >> ti.ds, wi = w >> ti.ds;
    *f = 2.0 * 2.0 * pc.cin * pc.cout * 9 * hi * wi;
    *by = (unfused ? 5.0 : 2.0) * pc.cin * hi * wi * 2;
    sig[1] = 31; sig[2] = pc.cin; sig[3] = pc.cout;
  } else if (op.kind == OP_BNECK) {
    // flops of the three convolutions; bytes: the launch reads x once and writes y once (unfused accounting: conv1 in + out,
    // conv2 in + out, conv3 in + residual + out)
    const TensorDesc& ti = net->tensors[op.in];
    const double px = (double)(h >> ti.ds) * (w >> ti.ds);
    const double cin = ti.C;   // 256: identity residual; 64: first Bottleneck, conv3 over [t2 ; x] (the projection of the residual)
    *f = 2.0 * px * (cin * 64 + 64.0 * 64 * 9 + 64.0 * 256 + (cin == 64 ? 64.0 * 256 : 0.0));
    *by = px * 2 * (unfused ? (cin + 64) + (64 + 64) + (64 + cin + 256) : cin + 256);
    sig[1] = 131; sig[2] = (int)cin; sig[3] = 256;
  } else if (op.kind == OP_FDOWN) {
    // fuse row 0 + the first down hops of branch 0: flops of the nb - 1 stride-2 convolutions; bytes: x0 in, y0 out, the
    // low-resolution terms in, the hops' outputs out (unfused accounting: every convolution and the sum read x0 for themselves)
    const TensorDesc& ti = net->tensors[op.in];
    const double hi = h >> ti.ds, wi = w >> ti.ds, c0 = ti.C;
    double cout = 0, low = 0;
    for (int k = 0; k < op.nouts; ++k) cout += net->tensors[op.outs[k]].C;
    for (int k = 0; k < op.nterms; ++k) low += c0 * (hi / (1 << op.shift[k])) * (wi / (1 << op.shift[k])) * 2;
    *f = 2.0 * c0 * cout * 9 * (hi / 2) * (wi / 2);
    *by = (unfused ? 2.0 + op.nouts : 2.0) * c0 * hi * wi * 2 + low + cout * (hi / 2) * (wi / 2) * 2;
    sig[1] = 32; sig[2] = (int)c0; sig[3] = (int)cout;
  } else if (op.kind == OP_HEAD) {
    const TensorDesc& ti = net->tensors[op.in];
    const double hi = h >> ti.ds, wi = w >> ti.ds, S = net->head_s, J = net->desc.num_joints;
    *by = ti.C * hi * wi * 2 + J * S * S * hi * wi * 4 + (op.res >= 0 ? J * S * S * hi * wi : 0);   // tap map + f32 out + coarser f32 level
    sig[1] = net->head_k * 10 + net->head_s; sig[2] = ti.C; sig[3] = net->desc.num_joints;
  } else {
    const TensorDesc& to = net->tensors[op.out];
    const double ho = h >> to.ds, wo = w >> to.ds;
    *by = to.C * ho * wo * 2;
    for (int k = 0; k < op.nterms; ++k) *by += to.C * (ho / (1 << op.shift[k])) * (wo / (1 << op.shift[k])) * 2;
    sig[1] = op.nterms; sig[2] = to.C; sig[3] = to.C;
  }
}

void hrnet_stats(scpose_hrnet* net, int h, int w, int* launches, double* flops, double* bytes) {
  double f = 0, by = 0;
  for (const Op& op : net->ops) {
    double of, ob; int32_t sig[4];
    op_work(net, op, h, w, &of, &ob, sig, true);   // whole-net figure in SURVEY.md 8(d)'s accounting (422 MB for W48 384^2)
    if (op.kind == OP_STEM || op.kind == OP_STEM2) ob -= 3.0 * h * w;   // network input is not an inter-layer activation
    f += of; by += ob;
  }
  int skipped = hrnet_tail_fused(net, 1, h, w) ? 1 : 0;   // the fused tail absorbs the last fuse row
  for (const scpose_hrnet::Chain& ch : net->chains)
    if (hrnet_chain_active(net, ch.first_op, h, w)) skipped += ch.nops - 1;   // a branch chain is one launch
  if (launches) *launches = (int)net->ops.size() - skipped;
  if (flops) *flops = f;
  if (bytes) *bytes = by;
}

void hrnet_free(scpose_hrnet* net) {
  for (auto& c : net->convs) conv_free(&c);
  if (net->d_stem_w) (void)hipFree(net->d_stem_w);
  if (net->d_stem_b) (void)hipFree(net->d_stem_b);
  if (net->d_mean_std) (void)hipFree(net->d_mean_std);
  for (auto& bn : net->bnecks) {
    if (bn.w1) (void)hipFree(bn.w1);
    if (bn.w2) (void)hipFree(bn.w2);
    if (bn.w3) (void)hipFree(bn.w3);
    if (bn.bias) (void)hipFree(bn.bias);
  }
  for (auto& fd : net->fdowns) fuse_down_free(&fd);
  for (auto& ch : net->chains) { if (ch.d_w) (void)hipFree(ch.d_w); if (ch.d_b) (void)hipFree(ch.d_b); }
  if (net->d_stemf_w1) (void)hipFree(net->d_stemf_w1);
  if (net->d_stemf_w2) (void)hipFree(net->d_stemf_w2);
  if (net->d_stemf_b1) (void)hipFree(net->d_stemf_b1);
  if (net->d_stemf_b2) (void)hipFree(net->d_stemf_b2);
  if (net->d_head_bias) (void)hipFree(net->d_head_bias);
  if (net->d_sched) (void)hipFree(net->d_sched);
  if (net->headf.d_w) (void)hipFree(net->headf.d_w);
  if (net->headf.d_b) (void)hipFree(net->headf.d_b);
  for (auto& e : net->events) if (e) (void)hipEventDestroy(e);
  net->events.clear();
}

}  // namespace scpose

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
using namespace scpose;

extern "C" int32_t scpose_hrnet_create(const scpose_hrnet_desc* desc, const char* const* names,
                                       const float* const* ptrs, const int64_t* numels,
                                       int32_t count, int32_t allow_missing, scpose_hrnet_t* out) {
  SCP_REQUIRE(desc && out, "hrnet_create: null argument");
  SCP_REQUIRE(desc->num_stages == 3, "hrnet_create: num_stages=%d (pose_hrnet has STAGE2..STAGE4)", desc->num_stages);
  SCP_REQUIRE(desc->num_joints > 0, "hrnet_create: num_joints=%d", desc->num_joints);
  SCP_REQUIRE(desc->final_conv_kernel == 1 || desc->final_conv_kernel == 3, "hrnet_create: FINAL_CONV_KERNEL=%d", desc->final_conv_kernel);
  SCP_REQUIRE(desc->dtype == SCPOSE_DT_BF16 || desc->dtype == SCPOSE_DT_F16, "hrnet_create: dtype=%d", desc->dtype);
  SCP_REQUIRE(desc->head >= SCPOSE_HEAD_FINAL_LAYER && desc->head <= SCPOSE_HEAD_CMS_384, "hrnet_create: head=%d", desc->head);
  if (desc->head != SCPOSE_HEAD_FINAL_LAYER) {
    SCP_REQUIRE(desc->final_conv_kernel == 1, "hrnet_create: the hrnet_cms heads are folded, which needs FINAL_CONV_KERNEL == 1 (got %d)", desc->final_conv_kernel);
    SCP_REQUIRE(desc->num_joints <= 16, "hrnet_create: the hrnet_cms heads support NUM_JOINTS <= 16 (got %d)", desc->num_joints);
  }
  for (int s = 0; s < 3; ++s) {
    SCP_REQUIRE(desc->block[s] == 0 || desc->block[s] == 1, "hrnet_create: STAGE%d BLOCK code %d (0 BASIC, 1 BOTTLENECK)", s + 2, desc->block[s]);
    SCP_REQUIRE(desc->num_branches[s] == s + 2, "hrnet_create: STAGE%d NUM_BRANCHES=%d (expected %d)", s + 2, desc->num_branches[s], s + 2);
    SCP_REQUIRE(desc->num_modules[s] >= 1, "hrnet_create: STAGE%d NUM_MODULES=%d", s + 2, desc->num_modules[s]);
    for (int b = 0; b < desc->num_branches[s]; ++b) {
      SCP_REQUIRE(desc->num_channels[s][b] > 0 && desc->num_channels[s][b] % 16 == 0,
                  "hrnet_create: STAGE%d NUM_CHANNELS[%d]=%d must be a positive multiple of 16", s + 2, b, desc->num_channels[s][b]);
      SCP_REQUIRE(desc->num_blocks[s][b] >= 1, "hrnet_create: STAGE%d NUM_BLOCKS[%d]=%d", s + 2, b, desc->num_blocks[s][b]);
      if (s > 0 && b < desc->num_branches[s - 1]) {   // channels a branch carries: planes * block.expansion
        const int was = desc->num_channels[s - 1][b] * (desc->block[s - 1] == 1 ? 4 : 1), is = desc->num_channels[s][b] * (desc->block[s] == 1 ? 4 : 1);
        SCP_REQUIRE(was == is,
                    "hrnet_create: STAGE%d branch %d changes channel count (%d -> %d); the reference forward (:445) cannot run that either",
                    s + 2, b, was, is);
      }
    }
  }
  Weights W;
  W.allow_missing = allow_missing != 0;
  for (int i = 0; i < count; ++i)
    if (names[i] && ptrs[i]) W.m[names[i]] = HostTensor{ptrs[i], numels[i]};
  scpose_hrnet* net = new (std::nothrow) scpose_hrnet();
  if (!net) { set_error("hrnet_create: out of host memory"); return SCPOSE_E_NOMEM; }
  net->desc = *desc;
  (void)hipGetDevice(&net->device);
  int32_t rc = hrnet_build(net, W);
  if (rc == SCPOSE_OK) {   // per-launch profiling events exist from create on: no launch function allocates anything
    net->events.assign(net->ops.size() + 1, nullptr);
    for (auto& e : net->events)
      if (hipEventCreate(&e) != hipSuccess) { set_error("hrnet_create: hipEventCreate failed"); rc = SCPOSE_E_HIP; break; }
  }
  if (rc != SCPOSE_OK) { hrnet_free(net); delete net; return rc; }
  *out = net;
  return SCPOSE_OK;
}

extern "C" int32_t scpose_hrnet_destroy(scpose_hrnet_t h) {
  if (!h) return SCPOSE_OK;
  hrnet_free(h);
  delete h;
  return SCPOSE_OK;
}

extern "C" int32_t scpose_hrnet_workspace_bytes(scpose_hrnet_t h, int32_t n, int32_t height,
                                                int32_t width, size_t* bytes) {
  SCP_REQUIRE(h && bytes, "hrnet_workspace_bytes: null argument");
  SCP_REQUIRE(n > 0 && height > 0 && width > 0 && height % 32 == 0 && width % 32 == 0,
              "hrnet_workspace_bytes: n=%d H=%d W=%d (H, W multiples of 32)", n, height, width);
  *bytes = hrnet_plan(h, n, height, width);
  return SCPOSE_OK;
}

extern "C" int32_t scpose_hrnet_heatmap_size(scpose_hrnet_t h, int32_t height, int32_t width, int32_t* out_h,
                                             int32_t* out_w) {
  SCP_REQUIRE(h && out_h && out_w, "hrnet_heatmap_size: null argument");
  SCP_REQUIRE(height > 0 && width > 0 && height % 32 == 0 && width % 32 == 0,
              "hrnet_heatmap_size: H=%d W=%d (multiples of 32)", height, width);
  *out_h = height / 4 * h->head_s;
  *out_w = width / 4 * h->head_s;
  return SCPOSE_OK;
}

extern "C" int32_t scpose_hrnet_stats(scpose_hrnet_t h, int32_t height, int32_t width,
                                      int32_t* launches, double* flops_per_frame,
                                      double* act_bytes_per_frame) {
  SCP_REQUIRE(h, "hrnet_stats: null handle");
  hrnet_stats(h, height, width, launches, flops_per_frame, act_bytes_per_frame);
  return SCPOSE_OK;
}

extern "C" int32_t scpose_hrnet_forward(scpose_hrnet_t h, const void* in, int32_t in_fmt, int32_t n,
                                        int32_t height, int32_t width, float* heatmaps,
                                        void* workspace, size_t workspace_bytes, void* stream) {
  SCP_REQUIRE(h && in && heatmaps, "hrnet_forward: null argument");
  return hrnet_forward(h, in, in_fmt, n, height, width, heatmaps, workspace, workspace_bytes,
                       static_cast<hipStream_t>(stream), false);
}

extern "C" int32_t scpose_hrnet_tail_fused(scpose_hrnet_t h, int32_t n, int32_t height, int32_t width, int32_t* fused) {
  SCP_REQUIRE(h && fused, "hrnet_tail_fused: null argument");
  SCP_REQUIRE(n > 0 && height > 0 && width > 0 && height % 32 == 0 && width % 32 == 0,
              "hrnet_tail_fused: n=%d H=%d W=%d (H, W multiples of 32)", n, height, width);
  *fused = hrnet_tail_fused(h, n, height, width) ? 1 : 0;
  return SCPOSE_OK;
}

extern "C" int32_t scpose_hrnet_forward_decode(scpose_hrnet_t h, const void* in, int32_t in_fmt, int32_t n, int32_t height,
                                               int32_t width, const float* center, const float* scale, int32_t post_process,
                                               float* preds_xyc, float* heatmaps, void* workspace, size_t workspace_bytes,
                                               void* stream) {
  SCP_REQUIRE(h && in && center && scale && preds_xyc, "hrnet_forward_decode: null argument");
  SCP_REQUIRE(h->desc.head == SCPOSE_HEAD_FINAL_LAYER || heatmaps, "hrnet_forward_decode: the hrnet_cms heads need a heat-map buffer");
  const scpose::HeadDecode dec{center, scale, post_process, preds_xyc};
  return hrnet_forward(h, in, in_fmt, n, height, width, heatmaps, workspace, workspace_bytes,
                       static_cast<hipStream_t>(stream), false, -1, 0, nullptr, &dec);
}

struct scpose_hrnet_graph {
  scpose_hrnet* net = nullptr;
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  hipStream_t cap = nullptr;
  scpose_hrnet_lanes lanes;
  int nodes = 0;
};

static void graph_free(scpose_hrnet_graph* g) {
  if (!g) return;
  if (g->exec) (void)hipGraphExecDestroy(g->exec);
  if (g->graph) (void)hipGraphDestroy(g->graph);
  for (auto& e : g->lanes.fork) if (e) (void)hipEventDestroy(e);
  for (auto& e : g->lanes.join) if (e) (void)hipEventDestroy(e);
  for (auto& st : g->lanes.side) if (st) (void)hipStreamDestroy(st);
  if (g->cap) (void)hipStreamDestroy(g->cap);
  delete g;
}

extern "C" int32_t scpose_hrnet_graph_workspace_bytes(scpose_hrnet_t h, int32_t n, int32_t height, int32_t width, size_t* bytes) {
  SCP_REQUIRE(h && bytes, "hrnet_graph_workspace_bytes: null argument");
  SCP_REQUIRE(n > 0 && height > 0 && width > 0 && height % 32 == 0 && width % 32 == 0,
              "hrnet_graph_workspace_bytes: n=%d H=%d W=%d (H, W multiples of 32)", n, height, width);
  *bytes = hrnet_plan(h, n, height, width, 1);
  return SCPOSE_OK;
}

static int32_t graph_create(scpose_hrnet_t h, const void* in, int32_t in_fmt, int32_t n, int32_t height,
                            int32_t width, float* heatmaps, void* workspace, size_t workspace_bytes,
                            int32_t concurrent, const scpose::HeadDecode* dec, scpose_hrnet_graph_t* out) {
  scpose_hrnet_graph* g = new (std::nothrow) scpose_hrnet_graph();
  if (!g) { set_error("hrnet_graph_create: out of host memory"); return SCPOSE_E_NOMEM; }
  g->net = h;
  auto fail = [&](int32_t rc) { graph_free(g); return rc; };
#define GC(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { set_error("%s failed: %s", #expr, hipGetErrorString(e_)); return fail(SCPOSE_E_HIP); } } while (0)
  GC(hipStreamCreateWithFlags(&g->cap, hipStreamNonBlocking));
  if (concurrent) {
    g->lanes.kinds = concurrent == 2 ? 0xcu : 0xeu;   // 2: only the fuse rows and transition convolutions run side by side
    int npar = 0, last = -1;
    for (const Op& op : h->ops) if (op.lane > 0 && op.epoch != last) { ++npar; last = op.epoch; }
    for (auto& st : g->lanes.side) GC(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    g->lanes.fork.assign(npar, nullptr);
    g->lanes.join.assign((size_t)npar * 3, nullptr);
    for (auto& e : g->lanes.fork) GC(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto& e : g->lanes.join) GC(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
  // One eager forward on this device first, through the SAME path the capture takes (same lanes, hence the same cu_share,
  // tilings and kernel instantiations per layer): per-device kernel attributes (hipFuncSetAttribute), the zero page and the
  // tile memos all exist before hipStreamBeginCapture, so the capture records nothing but launches (scpose.h contract).
  int32_t rc = hrnet_forward(h, in, in_fmt, n, height, width, heatmaps, workspace, workspace_bytes, g->cap, false, -1, 1,
                             concurrent ? &g->lanes : nullptr, dec);
  if (rc != SCPOSE_OK) return fail(rc);
  GC(hipStreamSynchronize(g->cap));   // the lanes were joined to g->cap at their epoch boundaries
  if (concurrent) for (auto& st : g->lanes.side) GC(hipStreamSynchronize(st));
  GC(hipStreamBeginCapture(g->cap, hipStreamCaptureModeThreadLocal));
  rc = hrnet_forward(h, in, in_fmt, n, height, width, heatmaps, workspace, workspace_bytes, g->cap, false, -1, 1,
                     concurrent ? &g->lanes : nullptr, dec);
  const hipError_t ce = hipStreamEndCapture(g->cap, &g->graph);
  if (rc != SCPOSE_OK) return fail(rc);
  if (ce != hipSuccess) { set_error("hipStreamEndCapture failed: %s", hipGetErrorString(ce)); return fail(SCPOSE_E_HIP); }
  GC(hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0));
  size_t nn = 0;
  if (hipGraphGetNodes(g->graph, nullptr, &nn) == hipSuccess) g->nodes = (int)nn;
#undef GC
  *out = g;
  return SCPOSE_OK;
}

extern "C" int32_t scpose_hrnet_graph_create(scpose_hrnet_t h, const void* in, int32_t in_fmt, int32_t n, int32_t height,
                                             int32_t width, float* heatmaps, void* workspace, size_t workspace_bytes,
                                             int32_t concurrent, scpose_hrnet_graph_t* out) {
  SCP_REQUIRE(h && in && heatmaps && workspace && out, "hrnet_graph_create: null argument");
  return graph_create(h, in, in_fmt, n, height, width, heatmaps, workspace, workspace_bytes, concurrent, nullptr, out);
}

extern "C" int32_t scpose_hrnet_graph_create_decode(scpose_hrnet_t h, const void* in, int32_t in_fmt, int32_t n, int32_t height,
                                                    int32_t width, const float* center, const float* scale, int32_t post_process,
                                                    float* preds_xyc, float* heatmaps, void* workspace, size_t workspace_bytes,
                                                    int32_t concurrent, scpose_hrnet_graph_t* out) {
  SCP_REQUIRE(h && in && center && scale && preds_xyc && workspace && out, "hrnet_graph_create_decode: null argument");
  SCP_REQUIRE(h->desc.head == SCPOSE_HEAD_FINAL_LAYER || heatmaps, "hrnet_graph_create_decode: the hrnet_cms heads need a heat-map buffer");
  const scpose::HeadDecode dec{center, scale, post_process, preds_xyc};
  return graph_create(h, in, in_fmt, n, height, width, heatmaps, workspace, workspace_bytes, concurrent, &dec, out);
}

extern "C" int32_t scpose_hrnet_graph_launch(scpose_hrnet_graph_t g, void* stream) {
  SCP_REQUIRE(g && g->exec, "hrnet_graph_launch: null graph");
  SCP_CHECK_HIP(hipGraphLaunch(g->exec, static_cast<hipStream_t>(stream)));
  return SCPOSE_OK;
}

extern "C" int32_t scpose_hrnet_graph_nodes(scpose_hrnet_graph_t g, int32_t* nodes) {
  SCP_REQUIRE(g && nodes, "hrnet_graph_nodes: null argument");
  *nodes = g->nodes;
  return SCPOSE_OK;
}

extern "C" int32_t scpose_hrnet_graph_destroy(scpose_hrnet_graph_t g) {
  graph_free(g);
  return SCPOSE_OK;
}

extern "C" int32_t scpose_hrnet_tap_names(scpose_hrnet_t h, char* buf, int32_t cap) {
  SCP_REQUIRE(h && buf && cap > 0, "hrnet_tap_names: null argument");
  std::string names;
  for (const auto& t : h->taps) names += (names.empty() ? "" : ",") + t.first;
  SCP_REQUIRE((int)names.size() < cap, "hrnet_tap_names: buffer of %d bytes too small (%zu needed)", cap, names.size() + 1);
  memcpy(buf, names.c_str(), names.size() + 1);
  return SCPOSE_OK;
}

extern "C" int32_t scpose_hrnet_forward_tap(scpose_hrnet_t h, const void* in, int32_t in_fmt, int32_t n, int32_t height,
                                            int32_t width, const char* tap, float* out, int32_t* channels, int32_t* out_h,
                                            int32_t* out_w, void* workspace, size_t workspace_bytes, void* stream) {
  SCP_REQUIRE(h && tap, "hrnet_forward_tap: null argument");
  int tid = -1;
  for (const auto& t : h->taps) if (t.first == tap) tid = t.second;
  if (tid < 0) {
    std::string names;
    for (const auto& t : h->taps) names += (names.empty() ? "" : ", ") + t.first;
    set_error("hrnet_forward_tap: unknown tap '%s' (available: %s)", tap, names.c_str());
    return SCPOSE_E_INVALID;
  }
  const TensorDesc& td = h->tensors[tid];
  if (channels) *channels = td.C;
  if (out_h) *out_h = height >> td.ds;
  if (out_w) *out_w = width >> td.ds;
  if (!out) return SCPOSE_OK;   // shape query
  SCP_REQUIRE(in, "hrnet_forward_tap: null input");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int32_t rc = hrnet_forward(h, in, in_fmt, n, height, width, nullptr, workspace, workspace_bytes, st, false, tid);
  if (rc != SCPOSE_OK) return rc;
  return blocked_to_nchw_launch(static_cast<const char*>(workspace) + h->plan[0].off[tid], n, td.C, height >> td.ds, width >> td.ds,
                                h->desc.dtype, out, st);
}

extern "C" int32_t scpose_hrnet_forward_profiled(scpose_hrnet_t h, const void* in, int32_t in_fmt,
                                                 int32_t n, int32_t height, int32_t width,
                                                 float* heatmaps, void* workspace,
                                                 size_t workspace_bytes, void* stream) {
  SCP_REQUIRE(h && in && heatmaps, "hrnet_forward_profiled: null argument");
  return hrnet_forward(h, in, in_fmt, n, height, width, heatmaps, workspace, workspace_bytes,
                       static_cast<hipStream_t>(stream), true);
}

extern "C" int32_t scpose_hrnet_forward_decode_profiled(scpose_hrnet_t h, const void* in, int32_t in_fmt, int32_t n, int32_t height,
                                                        int32_t width, const float* center, const float* scale, int32_t post_process,
                                                        float* preds_xyc, float* heatmaps, void* workspace, size_t workspace_bytes,
                                                        void* stream) {
  SCP_REQUIRE(h && in && center && scale && preds_xyc, "hrnet_forward_decode_profiled: null argument");
  SCP_REQUIRE(h->desc.head == SCPOSE_HEAD_FINAL_LAYER || heatmaps, "hrnet_forward_decode_profiled: the hrnet_cms heads need a heat-map buffer");
  const scpose::HeadDecode dec{center, scale, post_process, preds_xyc};
  return hrnet_forward(h, in, in_fmt, n, height, width, heatmaps, workspace, workspace_bytes,
                       static_cast<hipStream_t>(stream), true, -1, 0, nullptr, &dec);
}

extern "C" int32_t scpose_hrnet_profile_read(scpose_hrnet_t h, int32_t height, int32_t width,
                                             int32_t cap, float* ms, double* flops_per_frame,
                                             double* bytes_per_frame, int32_t* sig, int32_t* count) {
  SCP_REQUIRE(h && count, "hrnet_profile_read: null argument");
  const int nops = (int)h->ops.size();
  *count = nops;
  if (!ms) return SCPOSE_OK;   // size query
  SCP_REQUIRE(h->events_valid, "hrnet_profile_read: no profiled forward has been recorded");
  SCP_REQUIRE(cap >= nops, "hrnet_profile_read: capacity %d < %d ops", cap, nops);
  SCP_CHECK_HIP(hipEventSynchronize(h->events[nops]));
  for (int i = 0; i < nops; ++i) {
    SCP_CHECK_HIP(hipEventElapsedTime(&ms[i], h->events[i], h->events[i + 1]));
    double f, b; int32_t sg[4];
    op_work(h, h->ops[i], height, width, &f, &b, sg);
    if (flops_per_frame) flops_per_frame[i] = f;
    if (bytes_per_frame) bytes_per_frame[i] = b;
    if (sig) for (int k = 0; k < 4; ++k) sig[i * 4 + k] = sg[k];
  }
  return SCPOSE_OK;
}
