// hmgpu_api.hip -- host runtime of libhmgpu.so: context, device pictures (the DPB lives in HBM), staging of HM's
// per-CTU arrays, kernel sequencing, and the extern "C" surface declared in include/hmgpu.h.
//
// Host-side counterpart of TDecGop/TDecSlice/TDecCu's control flow (TDecGop.cpp:105-217), reduced to what is left
// once every traversal runs on the device: copy arrays, launch kernels, keep per-picture state.  The only serial
// host computation is reconstructBlkSAOParams' merge resolution (a dependent chain over CTUs, 3 x num_ctus items).
#include "hmgpu_dev.h"

#include <algorithm>
#include <chrono>
#include <atomic>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <new>
#include <vector>

using namespace hmgpu;

namespace {

enum { K_PREP = 0, K_MC_LUMA, K_MC_CHROMA, K_ITX, K_DBK_VER, K_DBK_HOR, K_SAO, K_EXTEND, K_H2D, K_INTRA, K_FILTER, K_RES11 };
const char* const kKernelNames[HMGPU_NUM_KERNELS] = {"prep", "mc_luma", "mc_chroma", "itx", "deblock_ver", "deblock_hor", "sao",
                                                     "extend_border", "h2d_stage", "intra", "filter_fused", ""};

struct SliceCall { int first_ctu, num_ctus, slice_idx; bool intra, wp, cells, bi, islice; };   // intra: the range holds intra CUs the device reconstructs; islice: mostly intra CUs;
                                                                                    // cells: it holds PUs that cut an 8x8 luma tile (k_mc_cells.hip); bi: B slices

struct Picture {
  bool in_use = false;
  bool sao_applied = false;
  bool filter_ready = false;            // SAO parameters staged
  bool sao_any = false;
  bool extended = false;                // margins of the final planes hold the replicated border
  std::vector<SliceCall> calls;
  // device allocations (owned)
  void* planes = nullptr;               // rec[3] + sao[3]
  void* meta = nullptr;                 // raw HM arrays
  void* coef = nullptr;
  void* pcm = nullptr;                  // PCM sample buffers, allocated when the first PCM CU shows up
  void* ccp = nullptr;                  // cross-component prediction weights (4:4:4), allocated with the first picture that carries them
  void* derived = nullptr;              // blk, tu lists, counters, sao params, slices
  uint8_t* sl_table = nullptr;          // device: expanded scaling-list matrices (inside `derived`)
  uint32_t* coef_start = nullptr;       // device: [3][num_ctus + 1] CTU starts of compact levels (inside `derived`)
  std::vector<uint8_t> sl_host;         // host copy the asynchronous upload reads from
  PicDev dev;                           // host mirror of the device descriptor
  std::vector<SliceDev> slices;         // host mirror of the slice table
  std::vector<SaoDev> h_saoprm;         // host copy of the resolved SAO parameters the asynchronous upload reads from
  std::vector<uint16_t> h_slice_idx, h_tile_idx;   // host mirrors of the per-CTU slice / tile index (SAO merge resolution needs them)
  int max_slice = -1;
  bool flags_staged = true;             // the device copies of transform_skip / bypass / ipcm may hold non-zero values
  uint64_t last_use = 0;                // use_seq of the last batch of kernels that read this picture's input arrays (0: none)
};

struct EventPair { hipEvent_t a, b; int kind; };

}  // namespace

// one page-locked block that holds a picture's input arrays in the order the device keeps them (hmgpu_staging_alloc)
struct hmgpu_staging {
  char* host = nullptr;
  size_t meta_bytes = 0, coef_bytes = 0, start_bytes = 0;   // metadata block | dense-capacity levels | [3][num_ctus + 1] CTU starts
  size_t grp[5] = {0, 0, 0, 0, 0};                         // carve_meta: where the optional groups of the metadata block start
  uint64_t copy_seq = 0;                                   // the staging pass (hmgpu_decompress_pictures) that last read the block ...
  hmgpu_ctx* reader = nullptr;                             // ... and the context it ran on (the owner, or one the block is shared with)
  hmgpu_ctx* owner = nullptr;
  std::vector<hmgpu_ctx*> sharers;                         // hmgpu_staging_share: contexts that take the block's arrays in one DMA too
  hmgpu_ctu_meta m;
  hmgpu_coeffs co;
};

struct hmgpu_ctx {
  hmgpu_seq_params seq;
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t stream2 = nullptr;      // second lane of the replay pipeline (hmgpu_set_streams)
  // host -> device staging of hmgpu_decompress_pictures runs on its own stream, so that the inputs of the next batch travel while the
  // kernels of this one run.  Two rings of events order it against the compute stream: copy_ev (inputs of a batch have arrived) and
  // use_ev (the kernels that read a picture's inputs have finished: its device arrays may be overwritten)
  hipStream_t copy_stream = nullptr;
  // ... and every second picture of a call on a second one: one stream's copies run on one DMA engine (~40 GB/s from page-locked memory on
  // this host), two reach 50-57 (tools/dbg/pcie.py).  The second lane joins the first before copy_ev is recorded (copy_join).
  hipStream_t copy_stream2 = nullptr;
  hipEvent_t copy_join = nullptr;
  hipEvent_t copy_ev[8] = {}, use_ev[8] = {};
  uint64_t copy_seq = 0, use_seq = 0;
  std::vector<hmgpu_staging*> stagings;
  std::vector<hmgpu_staging*> shared_stagings;   // blocks of other contexts (hmgpu_staging_share): recognised, not owned
  // Small host structures (descriptors, slice table entries, resolved SAO parameters) travel through a ring of page-locked memory: an
  // asynchronous copy from pageable memory makes the runtime stage the bytes itself, 20-100 us of the calling thread per copy (the resolved
  // SAO parameters of a picture: 0.18 ms; sixteen pictures per call spent 7 of their 7.7 ms on the host that way, round 4).  Eight segments; a
  // segment is reused when the events recorded at its close -- one per stream that may carry its copies -- have passed.
  static constexpr int kBounceSegs = 8;
  static constexpr size_t kBounceSeg = 1u << 20;
  char* bounce = nullptr;
  int bounce_seg = 0;
  size_t bounce_off = 0;
  hipEvent_t bounce_ev[kBounceSegs][3] = {};
  bool bounce_used[kBounceSegs] = {};
  // HMGPU_HOST_TIMING=1: wall time the calling thread spends inside the batch entry points, by part (printed by hmgpu_destroy)
  bool host_timing = false;
  double host_s[6] = {0, 0, 0, 0, 0, 0};
  uint64_t host_calls = 0;
  hipEvent_t dl_ev[32] = {};           // hmgpu_picture_download_begin tickets: ticket t completes with dl_ev[t % 32]
  std::atomic<uint64_t> dl_seq{0};
  // hmgpu_picture_hash_begin: MD5 chains of finished pictures over packed copies in a ring of slots; launched in batches (one lane per
  // plane, k_md5) on low-priority streams of their own
  static constexpr int kHashSlots = 96, kHashBatch = 32, kHashStreams = 1;   // (a batch: 96 chains = two waves; several streams shared hardware queues with each other and the context's own)
  hipStream_t hash_stream[kHashStreams] = {};
  hipEvent_t hash_packed[kHashSlots] = {}, hash_done[kHashSlots] = {};
  int hash_done_slot[kHashSlots] = {};   // the slot whose hash_done event stands for the batch a slot's chains ran in
  uint8_t* hash_buf[kHashSlots] = {};    // device: the packed planes, allocated when first used
  uint32_t* hash_dev = nullptr;          // device: [kHashSlots][12] state words
  uint32_t* hash_host = nullptr;         // page-locked: the same
  uint64_t hash_seq = 0, hash_launched = 0, hash_launches = 0;
  hipEvent_t xfer_ev[2] = {};            // hmgpu_picture_transfer: source ready / copy done
  uint64_t xfer_bytes = 0;
  uint32_t* dl_fault = nullptr;        // [32] page-locked: the picture's fault word (k_intra's bounded spin) as it stood behind the copies of ticket t
  std::vector<int> touched;            // pictures the entry point under way has enqueued work on, in any role (commit_use)
  std::vector<int> intra_launched;    // pictures whose intra kernel ran since the last fault check (k_intra's bounded spin)
  void* scratch = nullptr;            // device scratch of the output calls (packed download, picture hash): grown on demand, kept
  size_t scratch_bytes = 0;
  hipEvent_t lane_ev[2] = {nullptr, nullptr};
  int replay_streams = 1;
  int32_t last_err = 0;
  // geometry
  int ctu = 64, pw = 16, parts = 256, ctus_w = 0, ctus_h = 0, num_ctus = 0;
  int fmt = 1, csx = 1, csy = 1;          // chroma_format_idc (0 handled as 1: the chroma planes exist and are left alone) and its subsampling
  int pitch[3] = {0, 0, 0}, rows[3] = {0, 0, 0};
  int mx[3] = {0, 0, 0}, my[3] = {0, 0, 0};
  int grid_w = 0, grid_h = 0;
  uint32_t tu_cap[4] = {0, 0, 0, 0};
  size_t coef_elems[3] = {0, 0, 0};
  std::vector<Picture> pics;
  PicDev* d_pics = nullptr;
  PlaneSet* d_finals = nullptr;
  // sample planes of all device pictures in ONE allocation: picture i at plane_slab + i * 2 * plane_bytes (reconstruction planes, then
  // SAO planes), so that a kernel finds the final planes of a reference picture by arithmetic on its handle (McArgs, k_mc.hip)
  char* plane_slab = nullptr;
  size_t plane_bytes = 0;
  int32_t* d_ctu_order = nullptr;     // CTU addresses by anti-diagonal (dispatch order of the intra wavefront)
  std::vector<PlaneSet> h_finals;
  // profiling
  bool profiling = false;
  std::vector<EventPair> pending;
  std::vector<EventPair> free_events;
  double kernel_ms[HMGPU_NUM_KERNELS] = {0};
  uint64_t kernel_launches[HMGPU_NUM_KERNELS] = {0};
};

namespace {

struct HostTimer {                      // adds the time between construction and destruction to one slot (when timing is on)
  hmgpu_ctx* c; int slot; std::chrono::steady_clock::time_point t0;
  HostTimer(hmgpu_ctx* c_, int slot_) : c(c_), slot(slot_) { if (c->host_timing) t0 = std::chrono::steady_clock::now(); }
  ~HostTimer() { if (c->host_timing) c->host_s[slot] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
};

#define HIP_TRY(ctx, expr)                                   \
  do {                                                       \
    hipError_t e__ = (expr);                                 \
    if (e__ != hipSuccess) {                                 \
      (ctx)->last_err = (int32_t)e__;                        \
      return HMGPU_EDEVICE;                                  \
    }                                                        \
  } while (0)

// host -> device copy of a small structure on stream hs: through the context's page-locked ring (hmgpu_ctx::bounce) when it fits
static hipError_t h2d_small(hmgpu_ctx* c, void* dst, const void* src, size_t bytes, hipStream_t hs) {
  const size_t need = (bytes + 63) & ~(size_t)63;
  if (!c->bounce || need > hmgpu_ctx::kBounceSeg) return hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, hs);
  if (c->bounce_off + need > hmgpu_ctx::kBounceSeg) {
    hipStream_t streams[3] = {c->stream, c->copy_stream, c->copy_stream2};
    for (int k = 0; k < 3; k++) (void)hipEventRecord(c->bounce_ev[c->bounce_seg][k], streams[k]);
    c->bounce_used[c->bounce_seg] = true;
    c->bounce_seg = (c->bounce_seg + 1) % hmgpu_ctx::kBounceSegs;
    c->bounce_off = 0;
    if (c->bounce_used[c->bounce_seg]) for (int k = 0; k < 3; k++) (void)hipEventSynchronize(c->bounce_ev[c->bounce_seg][k]);
  }
  char* at = c->bounce + (size_t)c->bounce_seg * hmgpu_ctx::kBounceSeg + c->bounce_off;
  memcpy(at, src, bytes);
  c->bounce_off += need;
  return hipMemcpyAsync(dst, at, bytes, hipMemcpyHostToDevice, hs);
}


size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
// one set of planes of a picture: luma, then the plane that holds Cb and Cr
size_t plane_set_bytes(const hmgpu_ctx* c) {
  size_t n = 0;
  for (int k = 0; k < 2; k++) n += align_up((size_t)c->pitch[k] * c->rows[k] * sizeof(int16_t), 256);
  return n;
}

struct Carver {                       // sub-allocates one device block, 256-byte aligned pieces
  char* base; size_t off = 0;
  explicit Carver(void* b) : base((char*)b) {}
  template <typename T> T* take(size_t n) { T* p = base ? (T*)(base + off) : nullptr; off += align_up(n * sizeof(T), 256); return p; }
};

// device scratch of the output side: one allocation that lives with the context (hipMalloc / hipFree per call are device-wide
// synchronisations on the per-picture output path)
void* ctx_scratch(hmgpu_ctx* c, size_t bytes) {
  if (bytes > c->scratch_bytes) {
    if (c->scratch) { (void)hipStreamSynchronize(c->stream); (void)hipFree(c->scratch); c->scratch = nullptr; c->scratch_bytes = 0; }
    const size_t want = align_up(bytes + bytes / 4, 1 << 20);
    if (hipMalloc(&c->scratch, want) != hipSuccess) { c->scratch = nullptr; return nullptr; }
    c->scratch_bytes = want;
  }
  return c->scratch;
}

// after a synchronisation: did an intra wavefront give up waiting (k_intra.hip)?  The flag is sticky on the device until read here.
hmgpu_status check_faults(hmgpu_ctx* c) {
  hmgpu_status st = HMGPU_OK;
  for (int pic : c->intra_launched) {
    uint32_t f = 0;
    if (hipMemcpy(&f, c->pics[pic].dev.fault, 4, hipMemcpyDeviceToHost) != hipSuccess) { st = HMGPU_EDEVICE; continue; }
    if (f) { (void)hipMemset(c->pics[pic].dev.fault, 0, 4); c->last_err = -2; st = HMGPU_EDEVICE; }
  }
  c->intra_launched.clear();
  return st;
}

// Which submission last enqueued work on a picture, in ANY role (decoded, filtered, read as a reference, downloaded, hashed): the copy
// stream of hmgpu_decompress_pictures may rewrite a picture's descriptors and input arrays only behind that point.  Every entry point
// names the pictures it touches and ends with commit_use (one event on the context's stream).
void touch(hmgpu_ctx* c, int pic) { c->touched.push_back(pic); }
void commit_use(hmgpu_ctx* c) {
  if (c->touched.empty()) return;
  c->use_seq++;
  (void)hipEventRecord(c->use_ev[c->use_seq % 8], c->stream);
  for (int pic : c->touched) c->pics[pic].last_use = c->use_seq;
  c->touched.clear();
}
static void mark_use(hmgpu_ctx* c, const Batch& b) {
  for (int i = 0; i < b.n; i++) touch(c, b.pic[i]);
  commit_use(c);
}

// profiling: a pair of events around one launch, resolved lazily
void prof_begin(hmgpu_ctx* c, int kind, EventPair* ep) {
  if (!c->profiling) return;
  if (c->free_events.empty()) {
    hipEventCreate(&ep->a); hipEventCreate(&ep->b);
  } else { *ep = c->free_events.back(); c->free_events.pop_back(); }
  ep->kind = kind;
  hipEventRecord(ep->a, c->stream);
}
void prof_end(hmgpu_ctx* c, EventPair* ep) {
  if (!c->profiling) return;
  hipEventRecord(ep->b, c->stream);
  c->pending.push_back(*ep);
}
void prof_drain(hmgpu_ctx* c) {
  for (EventPair& ep : c->pending) {
    float ms = 0.f;
    hipEventSynchronize(ep.b);
    if (hipEventElapsedTime(&ms, ep.a, ep.b) == hipSuccess) { c->kernel_ms[ep.kind] += ms; c->kernel_launches[ep.kind]++; }
    c->free_events.push_back(ep);
  }
  c->pending.clear();
}
struct ProfScope {
  hmgpu_ctx* c; EventPair ep;
  ProfScope(hmgpu_ctx* ctx, int kind) : c(ctx) { prof_begin(c, kind, &ep); }
  ~ProfScope() { prof_end(c, &ep); if (c->pending.size() > 8192) prof_drain(c); }
};

// the raw HM arrays of one picture inside one block (device allocation; staging blocks mirror it, so that one copy moves them all).
// Order: what every picture needs first, then the groups a picture may do without -- list 1 (P slices), intra modes (no intra CUs),
// transform skip / lossless / PCM flags -- so that a copy from a staging block moves a prefix, or a prefix and one more piece.
// grp[0..4]: byte offsets where the groups start / the block ends.
void carve_meta(Carver& m, PicDev& d, size_t np, int num_ctus, size_t* grp = nullptr) {
  size_t g[5];
  g[0] = m.off;
  d.slice_idx = m.take<uint16_t>(num_ctus); d.tile_idx = m.take<uint16_t>(num_ctus);
  d.depth = m.take<uint8_t>(np); d.part_size = m.take<int8_t>(np); d.pred_mode = m.take<int8_t>(np);
  d.qp = m.take<int8_t>(np); d.tr_idx = m.take<uint8_t>(np);
  for (int k = 0; k < 3; k++) d.cbf[k] = m.take<uint8_t>(np);
  d.mv[0] = m.take<int16_t>(np * 2); d.ref_idx[0] = m.take<int8_t>(np);
  g[1] = m.off;
  d.mv[1] = m.take<int16_t>(np * 2); d.ref_idx[1] = m.take<int8_t>(np);
  g[2] = m.off;
  for (int k = 0; k < 2; k++) d.intra_dir[k] = m.take<uint8_t>(np);
  g[3] = m.off;
  for (int k = 0; k < 3; k++) d.tskip[k] = m.take<uint8_t>(np);
  d.bypass = m.take<uint8_t>(np); d.ipcm = m.take<uint8_t>(np);
  g[4] = m.off;
  if (grp) memcpy(grp, g, sizeof(g));
}

hmgpu_status alloc_picture(hmgpu_ctx* c, Picture& p) {
  const hmgpu_seq_params& s = c->seq;
  const size_t plane_bytes = c->plane_bytes;
  p.planes = c->plane_slab + (size_t)(&p - c->pics.data()) * 2 * plane_bytes;     // (zeroed with the slab)
  const size_t np = (size_t)c->num_ctus * c->parts;
  // raw metadata: 11 byte arrays + 2 mv arrays (4 B) + 2 ref_idx + slice/tile idx
  for (int pass = 0; pass < 2; pass++) {
    Carver m(pass ? p.meta : nullptr);
    carve_meta(m, p.dev, np, c->num_ctus);
    if (!pass) { HIP_TRY(c, hipMalloc(&p.meta, m.off)); HIP_TRY(c, hipMemset(p.meta, 0, m.off)); }
  }
  {
    size_t bytes = 0;
    for (int k = 0; k < 3; k++) bytes += align_up(c->coef_elems[k] * sizeof(int16_t), 256);
    HIP_TRY(c, hipMalloc(&p.coef, bytes));
    HIP_TRY(c, hipMemset(p.coef, 0, bytes));
    Carver m(p.coef);
    for (int k = 0; k < 3; k++) p.dev.coef[k] = m.take<int16_t>(c->coef_elems[k]);
  }
  for (int pass = 0; pass < 2; pass++) {
    Carver m(pass ? p.derived : nullptr);
    PicDev& d = p.dev;
    d.blk = m.take<BlkInfo>((size_t)c->grid_w * c->grid_h);
    d.edges = m.take<EdgeRec>((size_t)(c->grid_w / 2) * (c->grid_h / 2));
    d.tmv = m.take<TileMv>((size_t)(c->grid_w / 2) * (c->grid_h / 2));
    for (int k = 0; k < 3; k++) d.resid[k] = m.take<int16_t>(c->coef_elems[k]);
    for (int k = 0; k < 3; k++) d.quad_off[k] = m.take<uint32_t>((size_t)c->num_ctus * (c->parts / 4));
    p.coef_start = m.take<uint32_t>((size_t)3 * (c->num_ctus + 1));
    d.fault = m.take<uint32_t>(1);
    for (int k = 0; k < 4; k++) d.tu[k] = m.take<TuRec>((size_t)c->tu_cap[k] * kTuShards);
    d.tu_count = m.take<uint32_t>(4 * kTuShards);
    d.stats = m.take<unsigned long long>(2 * kTuShards);
    d.saoprm = m.take<SaoDev>((size_t)c->num_ctus * 3);
    d.slices = m.take<SliceDev>(HMGPU_MAX_SLICES);
    d.ctu_intra = m.take<uint8_t>((size_t)c->num_ctus);
    p.sl_table = m.take<uint8_t>(4 * 6 * 1024);
    d.intra_done = m.take<uint32_t>((size_t)3 * c->num_ctus);
    if (!pass) { HIP_TRY(c, hipMalloc(&p.derived, m.off)); HIP_TRY(c, hipMemset(p.derived, 0, m.off)); }
  }
  PicDev& d = p.dev;
  d.width = s.width; d.height = s.height;
  d.bd[0] = s.bit_depth_luma; d.bd[1] = d.bd[2] = s.bit_depth_chroma;
  d.log2ctu = s.log2_ctu_size; d.ctus_w = c->ctus_w; d.ctus_h = c->ctus_h; d.num_ctus = c->num_ctus; d.parts = c->parts; d.pw = c->pw;
  for (int k = 0; k < 3; k++) d.pitch[k] = c->pitch[k];
  d.grid_w = c->grid_w; d.grid_h = c->grid_h;
  d.lf_across_tiles = 1; d.sao_applied = 0;
  d.has_intra_dir = 0; d.strong_intra_smoothing = s.strong_intra_smoothing ? 1 : 0;
  d.range_ext = s.range_ext_flags;
  d.mono = s.chroma_format == 0 ? 1 : 0;
  d.fmt = c->fmt; d.csx = c->csx; d.csy = c->csy;
  d.ccp[0] = d.ccp[1] = nullptr;
  d.debug_skip_ctu = -1;
  d.sl_m = nullptr;
  for (int k = 0; k < 3; k++) { d.pcm[k] = nullptr; d.pcm_shift[k] = 0; d.coef_start[k] = nullptr; }
  d.pcm_lf_disable = s.pcm_loop_filter_disable ? 1 : 0; d.any_nofilt = 0;
  for (int k = 0; k < 4; k++) d.tu_cap[k] = c->tu_cap[k];
  {
    // plane pointers address sample (0,0); the margins lie at negative coordinates
    Carver m(p.planes);
    for (int k = 0; k < 3; k++) { d.mx[k] = c->mx[k]; d.my[k] = c->my[k]; }
    d.rec[0] = m.take<int16_t>((size_t)c->pitch[0] * c->rows[0]) + (size_t)c->my[0] * c->pitch[0] + c->mx[0];
    d.rec[1] = m.take<int16_t>((size_t)c->pitch[1] * c->rows[1]) + (size_t)c->my[1] * c->pitch[1] + kCStep * c->mx[1];
    d.rec[2] = d.rec[1] + 1;
    d.sao[0] = m.take<int16_t>((size_t)c->pitch[0] * c->rows[0]) + (size_t)c->my[0] * c->pitch[0] + c->mx[0];
    d.sao[1] = m.take<int16_t>((size_t)c->pitch[1] * c->rows[1]) + (size_t)c->my[1] * c->pitch[1] + kCStep * c->mx[1];
    d.sao[2] = d.sao[1] + 1;
  }
  p.slices.assign(HMGPU_MAX_SLICES, SliceDev());
  return HMGPU_OK;
}

void free_picture(Picture& p) {
  if (p.pcm) hipFree(p.pcm);                       // (the planes belong to the context's slab)
  if (p.ccp) hipFree(p.ccp);
  p.ccp = nullptr;
  if (p.meta) hipFree(p.meta);
  if (p.coef) hipFree(p.coef);
  if (p.derived) hipFree(p.derived);
  p.planes = p.meta = p.coef = p.derived = nullptr; p.pcm = nullptr;
}

hmgpu_status push_picdev(hmgpu_ctx* c, int pic) {
  HIP_TRY(c, h2d_small(c, c->d_pics + pic, &c->pics[pic].dev, sizeof(PicDev), c->stream));
  return HMGPU_OK;
}
hmgpu_status push_final(hmgpu_ctx* c, int pic) {
  Picture& p = c->pics[pic];
  for (int k = 0; k < 3; k++) c->h_finals[pic].p[k] = p.sao_applied ? p.dev.sao[k] : p.dev.rec[k];
  HIP_TRY(c, h2d_small(c, c->d_finals + pic, &c->h_finals[pic], sizeof(PlaneSet), c->stream));
  return HMGPU_OK;
}

bool valid_pic(const hmgpu_ctx* c, hmgpu_pic pic) { return pic >= 0 && pic < (int)c->pics.size() && c->pics[pic].in_use; }

// lazy border extension, as HM does when a picture first enters a reference list (TComSlice.cpp:350: extendPicBorder)
hmgpu_status ensure_extended(hmgpu_ctx* c, int pic) {
  Picture& p = c->pics[pic];
  touch(c, pic);                        // (called for every reference picture of a submission)
  if (p.extended) return HMGPU_OK;
  Batch b; memset(&b, 0, sizeof(b));
  b.n = 1; b.pic[0] = pic;
  { ProfScope ps(c, K_EXTEND); launch_extend(c->d_pics, b, c->seq.width, c->seq.height, c->mx[0], c->my[0], c->csx, c->csy, c->stream); }
  HIP_TRY(c, hipGetLastError());
  p.extended = true;
  return HMGPU_OK;
}
hmgpu_status ensure_refs_extended(hmgpu_ctx* c, const Batch& b, size_t call_idx) {
  for (int i = 0; i < b.n; i++) {
    const Picture& p = c->pics[b.pic[i]];
    if (call_idx >= p.calls.size()) continue;
    const SliceDev& sd = p.slices[p.calls[call_idx].slice_idx];
    for (int l = 0; l < 2; l++)
      for (int r = 0; r < HMGPU_MAX_REF; r++)
        if (sd.ref_pic[l][r] >= 0) { hmgpu_status st = ensure_extended(c, sd.ref_pic[l][r]); if (st != HMGPU_OK) return st; }
  }
  return HMGPU_OK;
}

// device work of one batch of slice calls (one call per picture): counters, prep, inverse transforms, MC (+ residual), intra
hmgpu_status run_recon(hmgpu_ctx* c, const Batch& b, bool any_intra, bool any_wp, bool any_cells, bool any_bi, bool any_islice) {
  int max_ctus = 0;
  for (int i = 0; i < b.n; i++) max_ctus = std::max(max_ctus, b.num_ctus[i]);
  // 4:2:2 / 4:4:4: the chroma of every inter cell comes from the format-generic kernel, which reads the BlkInfo grid
  { ProfScope ps(c, K_PREP); launch_prep(c->d_pics, b, max_ctus, c->parts, any_intra, any_cells || c->fmt != 1, c->fmt, c->stream); }
  if (c->fmt != 1) {
    // ... and adds the residual wherever it predicts: tiles no coded block covers must read as zero (Cb and Cr tiles are neighbours in memory)
    for (int i = 0; i < b.n; i++) {
      const PicDev& d = c->pics[b.pic[i]].dev;
      HIP_TRY(c, hipMemsetAsync(d.resid[1], 0, (size_t)((char*)d.resid[2] - (char*)d.resid[1]) + c->coef_elems[2] * sizeof(int16_t), c->stream));
    }
  }
  McArgs ma;
  memset(&ma, 0, sizeof(ma));
  ma.n = b.n; ma.width = c->seq.width; ma.height = c->seq.height; ma.log2ctu = c->seq.log2_ctu_size; ma.ctus_w = c->ctus_w;
  ma.tw = c->grid_w / 2; ma.npics = c->seq.max_pictures;
  ma.slab = c->plane_slab; ma.pic_stride = 2 * c->plane_bytes; ma.slab_bytes = ma.pic_stride * c->pics.size(); ma.sao_off = (uint32_t)c->plane_bytes;
  for (size_t i = 0; i < c->pics.size(); i++)
    if (c->pics[i].sao_applied) (i < 32 ? ma.sao_mask_lo : ma.sao_mask_hi) |= 1u << (i & 31);
  for (int i = 0; i < b.n; i++) {
    const PicDev& d = c->pics[b.pic[i]].dev;
    ma.first_ctu[i] = b.first_ctu[i]; ma.num_ctus[i] = b.num_ctus[i];
    ma.tmv[i] = d.tmv; ma.slices[i] = d.slices;
  }
  const PicDev& d0 = c->pics[b.pic[0]].dev;           // plane offsets inside a picture's part of the slab: the same for every picture
  const char* const base0 = (const char*)c->pics[b.pic[0]].planes;
  // the residual of the inter TUs first: the motion-compensation kernels add it when they write the prediction
  // workgroups per shard and size class: enough to keep the chip busy on one picture (768 per 2160p picture), fewer and longer-lived
  // ones when a batch of pictures fills it anyway (measured at 16 pictures: 12 -> 0.115 ms, 24 -> 0.122, 48 -> 0.135, 6 -> 0.120)
  uint32_t bps = (uint32_t)std::max(4, std::min(b.n >= 4 ? 12 : 24, max_ctus / 8 + 1));
  {
    ItxArgs ia;
    memset(&ia, 0, sizeof(ia));
    ia.n = b.n; ia.class_mask = 0xf;                        // all four size classes
    for (int k = 0; k < 3; k++) { ia.rtw[k] = (c->grid_w / 2) >> (k ? c->csx : 0); ia.bd[k] = d0.bd[k]; }
    ia.csx = c->csx; ia.csy = c->csy;
    for (int k = 0; k < 4; k++) ia.tu_cap[k] = c->tu_cap[k];
    for (int i = 0; i < b.n; i++) {
      const PicDev& d = c->pics[b.pic[i]].dev;
      for (int k = 0; k < 4; k++) ia.tu[i][k] = d.tu[k];
      ia.tu_count[i] = d.tu_count;
      for (int k = 0; k < 3; k++) { ia.coef[i][k] = d.coef[k]; ia.resid[i][k] = d.resid[k]; }
      ia.sl_m[i] = d.sl_m;
    }
    ProfScope ps(c, K_ITX);
    launch_itx(ia, bps, c->stream);
    bool any_ccp = false;
    for (int i = 0; i < b.n; i++) any_ccp |= c->pics[b.pic[i]].dev.ccp[0] != nullptr;
    if (any_ccp) launch_ccp(c->d_pics, b, max_ctus, c->stream);
  }
  {
    ProfScope ps(c, K_MC_LUMA);
    ma.pitch = c->pitch[0]; ma.bd = c->seq.bit_depth_luma;
    ma.origin_off = (uint32_t)((const char*)d0.rec[0] - base0);
    ma.rtw = c->grid_w / 2;
    for (int i = 0; i < b.n; i++) { ma.dst[i] = c->pics[b.pic[i]].dev.rec[0]; ma.resid[i] = c->pics[b.pic[i]].dev.resid[0]; }
    launch_mc_luma(ma, max_ctus, any_wp, any_bi, c->stream);
    if (any_cells) launch_mc_luma_cells(c->d_pics, c->d_finals, b, max_ctus, c->seq.log2_ctu_size, any_wp, c->stream);
  }
  if (c->fmt != 1) {
    ProfScope ps(c, K_MC_CHROMA);
    launch_mc_chroma_fmt(c->d_pics, c->d_finals, b, max_ctus, c->seq.log2_ctu_size, c->fmt, any_wp, c->stream);
  } else if (c->seq.chroma_format != 0) {
    ProfScope ps(c, K_MC_CHROMA);
    ma.pitch = c->pitch[1]; ma.bd = c->seq.bit_depth_chroma;
    ma.origin_off = (uint32_t)((const char*)d0.rec[1] - base0);      // the plane of both components (hmgpu_dev.h "chroma planes")
    ma.cr_off = 0;
    ma.rtw = c->grid_w / 4;
    for (int i = 0; i < b.n; i++) {
      const PicDev& d = c->pics[b.pic[i]].dev;
      ma.dst[i] = d.rec[1]; ma.dst2[i] = d.rec[2]; ma.resid[i] = d.resid[1]; ma.resid2[i] = d.resid[2];
    }
    launch_mc_chroma(ma, max_ctus, any_wp, any_bi, c->stream);
    if (any_cells) launch_mc_chroma_cells(c->d_pics, c->d_finals, b, max_ctus, c->seq.log2_ctu_size, any_wp, c->stream);
  }
  // intra CUs predict from finished neighbours (inter ones included): after motion compensation and the inter residuals
  if (any_intra) {
    ProfScope ps(c, K_INTRA);
    launch_intra(c->d_pics, b, c->d_ctu_order, c->num_ctus, !any_islice, c->stream);
    if (c->fmt == 2) launch_intra_chroma_422(c->d_pics, b, c->stream);        // (k_intra leaves the chroma of 4:2:2 pictures to it)
    for (int i = 0; i < b.n; i++) if (std::find(c->intra_launched.begin(), c->intra_launched.end(), b.pic[i]) == c->intra_launched.end()) c->intra_launched.push_back(b.pic[i]);
  }
  HIP_TRY(c, hipGetLastError());
  return HMGPU_OK;
}

hmgpu_status run_filter(hmgpu_ctx* c, const Batch& b, int stages) {
  // all three stages on pictures that all carry SAO: one pass through LDS instead of three through HBM (k_filter.hip)
  bool all_sao = stages == 7 && c->fmt == 1;            // (the fused kernel's tiles are those of 4:2:0 pictures)
  for (int i = 0; i < b.n && all_sao; i++) all_sao = c->pics[b.pic[i]].sao_any;
  if (all_sao) {
    bool nofilt = false;
    for (int i = 0; i < b.n; i++) nofilt |= c->pics[b.pic[i]].dev.any_nofilt != 0;
    { ProfScope ps(c, K_FILTER); launch_filter_fused(c->d_pics, b, c->seq.width, c->seq.height, nofilt, c->stream); }
    HIP_TRY(c, hipGetLastError());
    return HMGPU_OK;
  }
  // (4:2:2 / 4:4:4: k_deblock filters luma only, the chroma edges of the format's own grid follow from k_cfmt.hip -- per direction, as in HM)
  if (stages & 1) { ProfScope ps(c, K_DBK_VER); launch_deblock(c->d_pics, b, 0, c->seq.width, c->seq.height, c->stream);
                    if (c->fmt != 1) launch_deblock_chroma_fmt(c->d_pics, b, 0, c->seq.width, c->seq.height, c->stream); }
  if (stages & 2) { ProfScope ps(c, K_DBK_HOR); launch_deblock(c->d_pics, b, 1, c->seq.width, c->seq.height, c->stream);
                    if (c->fmt != 1) launch_deblock_chroma_fmt(c->d_pics, b, 1, c->seq.width, c->seq.height, c->stream); }
  if (stages & 4) {
    bool any = false;
    for (int i = 0; i < b.n; i++) any |= c->pics[b.pic[i]].sao_any;
    if (any) { ProfScope ps(c, K_SAO); launch_sao(c->d_pics, b, c->seq.width, c->seq.height, c->csx, c->csy, c->stream); }
  }
  HIP_TRY(c, hipGetLastError());
  return HMGPU_OK;
}

// reconstructBlkSAOParams (TComSampleAdaptiveOffset.cpp:229-372) + deriveLoopFilterBoundaryAvailibility
// (TComPicSym.cpp:365-471), host side: a dependent chain over CTUs, a few microseconds of work.
hmgpu_status stage_sao(hmgpu_ctx* c, Picture& p, const hmgpu_pic_params* pp, const hmgpu_sao_param* sao,
                       const std::vector<uint16_t>& slice_idx, const std::vector<uint16_t>& tile_idx) {
  const int n = c->num_ctus;
  // merge resolution by reference: res[a][comp] = the NEW / OFF entry that CTU a's parameters come from (a merged CTU takes all
  // three components of its left / upper neighbour's RESOLVED parameters); nothing of the caller's array is copied
  std::vector<const hmgpu_sao_param*> res((size_t)n * 3);
  std::vector<SaoDev>& dev = p.h_saoprm;  // lives with the picture: the upload below is asynchronous
  dev.resize((size_t)n * 3);
  bool any = false;
  for (int a = 0; a < n; a++) {
    const int cx = a % c->ctus_w, cy = a / c->ctus_w;
    const hmgpu_sao_param* const* merge[2] = {nullptr, nullptr};
    auto same = [&](int o) { return slice_idx[o] == slice_idx[a] && tile_idx[o] == tile_idx[a]; };
    if (cx > 0 && same(a - 1)) merge[HMGPU_SAO_MERGE_LEFT] = &res[(size_t)(a - 1) * 3];
    if (cy > 0 && same(a - c->ctus_w)) merge[HMGPU_SAO_MERGE_ABOVE] = &res[(size_t)(a - c->ctus_w) * 3];
    // neighbour availability as a 3x3 grid (SaoDev::avail): bit 3 * (dy + 1) + (dx + 1)
    unsigned avail = 1u << 4;
    for (int k = 0; k < 9; k++) {
      if (k == 4) continue;
      const int nx = cx + k % 3 - 1, ny = cy + k / 3 - 1;
      if (nx < 0 || nx >= c->ctus_w || ny < 0 || ny >= c->ctus_h) continue;
      const int o = ny * c->ctus_w + nx;
      bool ok = true;
      if (slice_idx[o] != slice_idx[a]) {
        // the slice that comes later in decoding order decides with its own flag (TComPicSym.cpp:403-452)
        const SliceDev& later = p.slices[std::max(slice_idx[o], slice_idx[a])];
        ok = later.lf_across_slices != 0;
      }
      if (ok && !pp->lf_across_tiles) ok = tile_idx[o] == tile_idx[a];
      if (ok) avail |= 1u << k;
    }
    for (int comp = 0; comp < 3; comp++) {
      const hmgpu_sao_param* r = &sao[(size_t)a * 3 + comp];
      if (r->mode_idc == HMGPU_SAO_MERGE) {
        if (r->type_idc < 0 || r->type_idc > 1 || !merge[r->type_idc]) return HMGPU_EINVAL;     // HM: assert(mergeTarget != NULL)
        r = merge[r->type_idc][comp];
      }
      res[(size_t)a * 3 + comp] = r;
      const int shift = comp == 0 ? pp->sao_offset_shift_luma : pp->sao_offset_shift_chroma;
      SaoDev& d = dev[(size_t)a * 3 + comp];
      memset(&d, 0, sizeof(d));
      d.type = r->mode_idc == HMGPU_SAO_OFF ? -1 : (int8_t)r->type_idc;
      d.avail = (uint16_t)avail;
      // offsets of a NEW entry: the coded ones scaled by log2_sao_offset_scale (reconstructBlkSAOParam, TComSampleAdaptiveOffset.cpp:229-372)
      if (d.type == HMGPU_SAO_BO) {
        d.band = (uint8_t)(r->type_aux_info & 31);
        for (int i = 0; i < 4; i++) d.off[i] = (int8_t)(r->offset[(r->type_aux_info + i) & 31] * (1 << shift));
      } else if (d.type >= 0) {
        for (int i = 0; i < 5; i++) d.off[i] = (int8_t)(r->offset[i] * (1 << shift));
      }
      any |= d.type >= 0;
    }
  }
  p.sao_any = any;
  HIP_TRY(c, h2d_small(c, p.dev.saoprm, dev.data(), dev.size() * sizeof(SaoDev), c->stream));
  return HMGPU_OK;
}

}  // namespace

// ======================================================================================================= C ABI
extern "C" {

const char* hmgpu_status_string(hmgpu_status s) {
  switch (s) {
    case HMGPU_OK: return "ok";
    case HMGPU_EINVAL: return "invalid argument";
    case HMGPU_EDEVICE: return "device (HIP) error";
    case HMGPU_EUNSUPPORTED: return "coding tool not supported";
    case HMGPU_ENOMEM: return "out of memory";
  }
  return "?";
}
const char* hmgpu_kernel_name(int32_t k) { return (k >= 0 && k < HMGPU_NUM_KERNELS) ? kKernelNames[k] : ""; }

int32_t hmgpu_num_ctus(const hmgpu_seq_params* seq) {
  const int c = 1 << seq->log2_ctu_size;
  return ((seq->width + c - 1) / c) * ((seq->height + c - 1) / c);
}
int32_t hmgpu_parts_per_ctu(const hmgpu_seq_params* seq) { return 1 << (2 * seq->log2_ctu_size - 4); }

hmgpu_status hmgpu_create(const hmgpu_seq_params* seq, int device_ordinal, hmgpu_ctx** out) {
  if (!seq || !out) return HMGPU_EINVAL;
  *out = nullptr;
  if (seq->width <= 0 || seq->height <= 0 || (seq->width & 7) || (seq->height & 7)) return HMGPU_EINVAL;
  if (seq->log2_ctu_size < 4 || seq->log2_ctu_size > 6) return HMGPU_EINVAL;
  if (seq->max_pictures < 1 || seq->max_pictures > kMaxPics) return HMGPU_EINVAL;
  if (seq->chroma_format < 0 || seq->chroma_format > 3) return HMGPU_EINVAL;              // 0: monochrome -- the chroma planes exist and are left alone
  if (seq->range_ext_flags & ~(HMGPU_REXT_ROTATION | HMGPU_REXT_IMPLICIT_RDPCM | HMGPU_REXT_EXPLICIT_RDPCM | HMGPU_REXT_INTRA_SMOOTHING_DISABLED)) return HMGPU_EUNSUPPORTED;
  if (seq->bit_depth_luma < 8 || seq->bit_depth_luma > 12 || seq->bit_depth_chroma < 8 || seq->bit_depth_chroma > 12) return HMGPU_EUNSUPPORTED;
  hmgpu_ctx* c = new (std::nothrow) hmgpu_ctx();
  if (!c) return HMGPU_ENOMEM;
  c->seq = *seq;
  c->device = device_ordinal;
  c->host_timing = getenv("HMGPU_HOST_TIMING") != nullptr;
  hipError_t e = hipSetDevice(device_ordinal);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->copy_stream2, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->copy_join, hipEventDisableTiming);
  if (e == hipSuccess) e = hipHostMalloc((void**)&c->bounce, hmgpu_ctx::kBounceSegs * hmgpu_ctx::kBounceSeg, hipHostMallocDefault);
  for (int g = 0; g < hmgpu_ctx::kBounceSegs && e == hipSuccess; g++)
    for (int k = 0; k < 3 && e == hipSuccess; k++) e = hipEventCreateWithFlags(&c->bounce_ev[g][k], hipEventDisableTiming);
  for (int k = 0; k < 8 && e == hipSuccess; k++) e = hipEventCreateWithFlags(&c->copy_ev[k], hipEventDisableTiming);
  for (int k = 0; k < 8 && e == hipSuccess; k++) e = hipEventCreateWithFlags(&c->use_ev[k], hipEventDisableTiming);
  for (int k = 0; k < 32 && e == hipSuccess; k++) e = hipEventCreateWithFlags(&c->dl_ev[k], hipEventDisableTiming | hipEventBlockingSync);   // (waited for by helper threads: sleep, do not spin)
  if (e == hipSuccess) { e = hipHostMalloc((void**)&c->dl_fault, 32 * sizeof(uint32_t), hipHostMallocDefault); if (e == hipSuccess) memset(c->dl_fault, 0, 32 * sizeof(uint32_t)); }
  for (int k = 0; k < 2 && e == hipSuccess; k++) e = hipEventCreateWithFlags(&c->lane_ev[k], hipEventDisableTiming);
  if (e != hipSuccess) { delete c; return HMGPU_EDEVICE; }
  c->ctu = 1 << seq->log2_ctu_size; c->pw = c->ctu / 4; c->parts = c->pw * c->pw;
  c->ctus_w = (seq->width + c->ctu - 1) / c->ctu; c->ctus_h = (seq->height + c->ctu - 1) / c->ctu;
  c->num_ctus = c->ctus_w * c->ctus_h;
  c->grid_w = c->ctus_w * c->pw; c->grid_h = c->ctus_h * c->pw;
  // planes: HM's picture-buffer shape (TComPicYuv.cpp:89-100: margins of maxCU + 16 luma samples all around, extended by
  // replication) with device-friendly numbers: 128-sample (256-byte) horizontal margins keep sample (0,y) cache-line
  // aligned, rows are a multiple of 128 bytes plus one spare line (vector loads may run past the margin; the pitch is
  // never a power of two)
  c->fmt = seq->chroma_format == 0 ? 1 : seq->chroma_format;
  c->csx = c->fmt == 3 ? 0 : 1; c->csy = c->fmt == 1 ? 1 : 0;
  c->mx[0] = 128; c->mx[1] = c->mx[2] = 128 >> c->csx;
  c->my[0] = 80; c->my[1] = c->my[2] = 80 >> c->csy;
  c->pitch[0] = (int)align_up((size_t)seq->width, 64) + 2 * c->mx[0] + 64;
  // (Cb and Cr alternate in one plane, hmgpu_dev.h "chroma planes": its pitch is that of kCStep rows of one component)
  c->pitch[1] = c->pitch[2] = kCStep * ((int)align_up((size_t)seq->width >> c->csx, 64) + 2 * c->mx[1] + 64);
  c->rows[0] = c->ctus_h * c->ctu + 2 * c->my[0] + 8; c->rows[1] = c->rows[2] = ((c->ctus_h * c->ctu) >> c->csy) + 2 * c->my[1] + 8;
  c->coef_elems[0] = (size_t)c->num_ctus * c->ctu * c->ctu;
  c->coef_elems[1] = c->coef_elems[2] = c->coef_elems[0] >> (c->csx + c->csy);
  {
    // TU list capacity of one shard: prep blocks (256 threads x 4 partitions = 1024 partitions) go round-robin to the
    // shards; 256 partitions (one 64x64 luma area) hold at most 256+128 4x4, 64+32 8x8, 16+8 16x16 and 4 32x32 TUs
    const size_t blocks = ((size_t)c->num_ctus * (c->parts / 4) + 255) / 256;
    const size_t per_shard = (blocks + kTuShards - 1) / kTuShards;
    const uint32_t per_block[4] = {4 * 384, 4 * 96, 4 * 24, 4 * 4};
    // (4:2:2 / 4:4:4: up to as many chroma blocks per component as luma blocks, of every size)
    for (int k = 0; k < 4; k++) c->tu_cap[k] = (uint32_t)(per_shard * (c->fmt == 1 ? per_block[k] : 3u * (1024u >> (2 * k))));
  }
  c->pics.resize(seq->max_pictures);
  c->h_finals.resize(seq->max_pictures);
  hmgpu_status st = HMGPU_OK;
  c->plane_bytes = plane_set_bytes(c);
  if (hipMalloc((void**)&c->plane_slab, c->plane_bytes * 2 * seq->max_pictures) != hipSuccess ||
      hipMemset(c->plane_slab, 0, c->plane_bytes * 2 * seq->max_pictures) != hipSuccess) st = HMGPU_EDEVICE;
  for (int i = 0; i < seq->max_pictures && st == HMGPU_OK; i++) st = alloc_picture(c, c->pics[i]);
  if (st == HMGPU_OK) {
    {
      std::vector<int32_t> order;
      for (int d = 0; d <= 2 * (c->ctus_h - 1) + c->ctus_w - 1; d++)
        for (int r = 0; r < c->ctus_h; r++) { const int col = d - 2 * r; if (col >= 0 && col < c->ctus_w) order.push_back(r * c->ctus_w + col); }
      if (hipMalloc((void**)&c->d_ctu_order, order.size() * sizeof(int32_t)) != hipSuccess ||
          hipMemcpy(c->d_ctu_order, order.data(), order.size() * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess) st = HMGPU_EDEVICE;
    }
    if (hipMalloc((void**)&c->d_pics, sizeof(PicDev) * seq->max_pictures) != hipSuccess ||
        hipMalloc((void**)&c->d_finals, sizeof(PlaneSet) * seq->max_pictures) != hipSuccess) st = HMGPU_EDEVICE;
  }
  if (st == HMGPU_OK) {
    for (int i = 0; i < seq->max_pictures && st == HMGPU_OK; i++) { st = push_picdev(c, i); if (st == HMGPU_OK) st = push_final(c, i); }
    if (st == HMGPU_OK && hipStreamSynchronize(c->stream) != hipSuccess) st = HMGPU_EDEVICE;
  }
  if (st != HMGPU_OK) { hmgpu_destroy(c); return st; }
  *out = c;
  return HMGPU_OK;
}

void hmgpu_destroy(hmgpu_ctx* c) {
  if (!c) return;
  hipSetDevice(c->device);
  if (c->stream) hipStreamSynchronize(c->stream);
  prof_drain(c);
  for (EventPair& ep : c->free_events) { hipEventDestroy(ep.a); hipEventDestroy(ep.b); }
  if (c->host_timing && c->host_calls)
    fprintf(stderr, "hmgpu host time per hmgpu_decompress_pictures + hmgpu_filter_pictures (%llu calls): validate %.3f ms, slices %.3f, stage_inputs %.3f, recon launches %.3f, sao staging %.3f, filter launches %.3f\n",
            (unsigned long long)c->host_calls, 1e3 * c->host_s[0] / c->host_calls, 1e3 * c->host_s[1] / c->host_calls, 1e3 * c->host_s[2] / c->host_calls,
            1e3 * c->host_s[3] / c->host_calls, 1e3 * c->host_s[4] / c->host_calls, 1e3 * c->host_s[5] / c->host_calls);
  for (Picture& p : c->pics) free_picture(p);
  if (c->d_pics) hipFree(c->d_pics);
  if (c->d_finals) hipFree(c->d_finals);
  if (c->plane_slab) hipFree(c->plane_slab);
  if (c->scratch) hipFree(c->scratch);
  if (c->d_ctu_order) hipFree(c->d_ctu_order);
  if (c->stream) hipStreamDestroy(c->stream);
  if (c->stream2) hipStreamDestroy(c->stream2);
  if (c->copy_stream2) { hipStreamSynchronize(c->copy_stream2); hipStreamDestroy(c->copy_stream2); }
  if (c->copy_stream) { hipStreamSynchronize(c->copy_stream); hipStreamDestroy(c->copy_stream); }
  if (c->copy_join) hipEventDestroy(c->copy_join);
  for (int g = 0; g < hmgpu_ctx::kBounceSegs; g++) for (int k = 0; k < 3; k++) if (c->bounce_ev[g][k]) hipEventDestroy(c->bounce_ev[g][k]);
  if (c->bounce) (void)hipHostFree(c->bounce);
  for (int k = 0; k < 8; k++) { if (c->copy_ev[k]) hipEventDestroy(c->copy_ev[k]); if (c->use_ev[k]) hipEventDestroy(c->use_ev[k]); }
  for (hmgpu_staging* st : c->shared_stagings) {
    st->sharers.erase(std::remove(st->sharers.begin(), st->sharers.end(), c), st->sharers.end());
    if (st->reader == c) { st->reader = nullptr; st->copy_seq = 0; }      // (the copy stream was drained above)
  }
  for (hmgpu_staging* st : c->stagings) {
    for (hmgpu_ctx* o : st->sharers) o->shared_stagings.erase(std::remove(o->shared_stagings.begin(), o->shared_stagings.end(), st), o->shared_stagings.end());
    if (st->host) hipHostFree(st->host);
    delete st;
  }
  for (int k = 0; k < 32; k++) if (c->dl_ev[k]) hipEventDestroy(c->dl_ev[k]);
  if (c->dl_fault) (void)hipHostFree(c->dl_fault);
  for (int k = 0; k < hmgpu_ctx::kHashStreams; k++) if (c->hash_stream[k]) { (void)hipStreamSynchronize(c->hash_stream[k]); (void)hipStreamDestroy(c->hash_stream[k]); }
  for (int k = 0; k < hmgpu_ctx::kHashSlots; k++) {
    if (c->hash_packed[k]) hipEventDestroy(c->hash_packed[k]);
    if (c->hash_done[k]) hipEventDestroy(c->hash_done[k]);
    if (c->hash_buf[k]) (void)hipFree(c->hash_buf[k]);
  }
  if (c->hash_dev) (void)hipFree(c->hash_dev);
  for (int k = 0; k < 2; k++) if (c->xfer_ev[k]) hipEventDestroy(c->xfer_ev[k]);
  if (c->hash_host) (void)hipHostFree(c->hash_host);
  for (int k = 0; k < 2; k++) if (c->lane_ev[k]) hipEventDestroy(c->lane_ev[k]);
  delete c;
}

int32_t hmgpu_last_device_error(const hmgpu_ctx* c) { return c ? c->last_err : 0; }

hmgpu_status hmgpu_debug_stall_intra(hmgpu_ctx* c, hmgpu_pic pic, int32_t ctu) {
  if (!c || !valid_pic(c, pic) || ctu < -1 || ctu >= c->num_ctus) return HMGPU_EINVAL;
  c->pics[pic].dev.debug_skip_ctu = ctu;               // (reaches the device with the picture's next decompress call)
  return HMGPU_OK;
}

void* hmgpu_host_alloc(size_t bytes) {
  void* p = nullptr;
  if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  return p;
}
void hmgpu_host_free(void* p) { if (p) (void)hipHostFree(p); }

hmgpu_status hmgpu_sync(hmgpu_ctx* c) {
  if (c && c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
  if (!c) return HMGPU_EINVAL;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  prof_drain(c);
  return check_faults(c);
}

hmgpu_status hmgpu_picture_acquire(hmgpu_ctx* c, hmgpu_pic* out) {
  if (!c || !out) return HMGPU_EINVAL;
  for (size_t i = 0; i < c->pics.size(); i++) {
    Picture& p = c->pics[i];
    if (!p.in_use) {
      p.in_use = true; p.sao_applied = false; p.filter_ready = false; p.sao_any = false; p.calls.clear(); p.max_slice = -1;
      p.extended = false;
      p.dev.sao_applied = 0; p.dev.any_nofilt = 0;
      *out = (hmgpu_pic)i;
      hmgpu_status st = push_final(c, (int)i);
      return st;
    }
  }
  return HMGPU_ENOMEM;
}

hmgpu_status hmgpu_picture_release(hmgpu_ctx* c, hmgpu_pic pic) {
  if (!c || !valid_pic(c, pic)) return HMGPU_EINVAL;
  c->pics[pic].in_use = false;
  return HMGPU_OK;
}

hmgpu_status hmgpu_picture_upload(hmgpu_ctx* c, hmgpu_pic pic, const int16_t* const planes[3], const int32_t strides[3]) {
  if (!c || !valid_pic(c, pic) || !planes || !strides) return HMGPU_EINVAL;
  Picture& p = c->pics[pic];
  if (p.sao_applied) {                 // uploaded samples ARE the picture: back to the reconstruction planes
    p.sao_applied = false; p.dev.sao_applied = 0;
    hmgpu_status st = push_final(c, pic);
    if (st == HMGPU_OK) st = push_picdev(c, pic);
    if (st != HMGPU_OK) return st;
  }
  HIP_TRY(c, hipMemcpy2DAsync(p.dev.rec[0], (size_t)c->pitch[0] * 2, planes[0], (size_t)strides[0] * 2, (size_t)c->seq.width * 2, c->seq.height,
                              hipMemcpyHostToDevice, c->stream));
  {
    // the chroma components arrive as HM's two planes and are laid sample by sample into the device's one (hmgpu_dev.h "chroma planes")
    const int w = c->seq.width >> c->csx, h = c->seq.height >> c->csy;
    int16_t* d = static_cast<int16_t*>(ctx_scratch(c, (size_t)2 * w * h * sizeof(int16_t)));
    if (!d) return HMGPU_ENOMEM;
    for (int k = 1; k < 3; k++) {
      int16_t* dk = d + (size_t)(k - 1) * w * h;
      HIP_TRY(c, hipMemcpy2DAsync(dk, (size_t)w * 2, planes[k], (size_t)strides[k] * 2, (size_t)w * 2, h, hipMemcpyHostToDevice, c->stream));
      launch_unpack(dk, w, h, p.dev.rec[k], c->pitch[k], kCStep, c->stream);
    }
    HIP_TRY(c, hipGetLastError());
  }
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  p.extended = false;
  return HMGPU_OK;
}

// the final planes of a picture to HM's three host planes, enqueued on the context's stream: luma straight from its plane, the chroma
// components taken apart into a dense block of device scratch first (later users of the scratch follow on the same stream)
static hmgpu_status enqueue_download(hmgpu_ctx* c, Picture& p, int16_t* const planes[3], const int32_t strides[3]) {
  const int16_t* y = p.sao_applied ? p.dev.sao[0] : p.dev.rec[0];
  HIP_TRY(c, hipMemcpy2DAsync(planes[0], (size_t)strides[0] * 2, y, (size_t)c->pitch[0] * 2, (size_t)c->seq.width * 2, c->seq.height, hipMemcpyDeviceToHost, c->stream));
  const int w = c->seq.width >> c->csx, h = c->seq.height >> c->csy;
  int16_t* d = static_cast<int16_t*>(ctx_scratch(c, (size_t)2 * w * h * sizeof(int16_t)));
  if (!d) return HMGPU_ENOMEM;
  for (int k = 1; k < 3; k++) {
    const int16_t* src = p.sao_applied ? p.dev.sao[k] : p.dev.rec[k];
    int16_t* dk = d + (size_t)(k - 1) * w * h;
    launch_pack(src, c->pitch[k], kCStep, 0, 0, w, h, 2, reinterpret_cast<uint8_t*>(dk), w * 2, c->stream);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpy2DAsync(planes[k], (size_t)strides[k] * 2, dk, (size_t)w * 2, (size_t)w * 2, h, hipMemcpyDeviceToHost, c->stream));
  }
  return HMGPU_OK;
}

hmgpu_status hmgpu_picture_download(hmgpu_ctx* c, hmgpu_pic pic, int16_t* const planes[3], const int32_t strides[3]) {
  if (!c || !valid_pic(c, pic) || !planes || !strides) return HMGPU_EINVAL;
  Picture& p = c->pics[pic];
  { const hmgpu_status st = enqueue_download(c, p, planes, strides); if (st != HMGPU_OK) return st; }
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  prof_drain(c);
  return check_faults(c);
}

hmgpu_status hmgpu_picture_download_begin(hmgpu_ctx* c, hmgpu_pic pic, int16_t* const planes[3], const int32_t strides[3], uint64_t* ticket) {
  if (!c || !valid_pic(c, pic) || !planes || !strides || !ticket) return HMGPU_EINVAL;
  hipSetDevice(c->device);
  Picture& p = c->pics[pic];
  { const hmgpu_status st = enqueue_download(c, p, planes, strides); if (st != HMGPU_OK) return st; }
  const uint64_t t = c->dl_seq.load() + 1;
  // the picture's fault word (an intra wavefront that gave up waiting) as it stands behind these copies: hmgpu_download_wait reports it
  HIP_TRY(c, hipMemcpyAsync(&c->dl_fault[t % 32], p.dev.fault, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipEventRecord(c->dl_ev[t % 32], c->stream));
  c->dl_seq.store(t);
  *ticket = t;
  touch(c, pic);
  commit_use(c);
  return HMGPU_OK;
}

hmgpu_status hmgpu_download_wait(hmgpu_ctx* c, uint64_t ticket) {
  if (!c || ticket == 0 || ticket > c->dl_seq.load()) return HMGPU_EINVAL;
  hipSetDevice(c->device);
  // a ticket whose event has been re-recorded is 32 downloads old: the event now stands for a LATER point of the same stream
  if (hipEventSynchronize(c->dl_ev[ticket % 32]) != hipSuccess) return HMGPU_EDEVICE;
  // (read-only here: this may be a helper thread; the word on the device stays set until the context's own thread reads it in check_faults)
  if (c->dl_seq.load() - ticket < 32 && reinterpret_cast<volatile uint32_t*>(c->dl_fault)[ticket % 32]) return HMGPU_EDEVICE;
  return HMGPU_OK;
}

hmgpu_status hmgpu_picture_download_packed(hmgpu_ctx* c, hmgpu_pic pic, void* const planes[3], const int32_t stride_bytes[3],
                                           int32_t bytes_per_sample, int32_t crop_left, int32_t crop_right, int32_t crop_top, int32_t crop_bottom) {
  if (!c || !valid_pic(c, pic) || !planes || !stride_bytes || (bytes_per_sample != 1 && bytes_per_sample != 2)) return HMGPU_EINVAL;
  if (((crop_left | crop_right) & ((1 << c->csx) - 1)) || ((crop_top | crop_bottom) & ((1 << c->csy) - 1))) return HMGPU_EINVAL;   // whole chroma samples
  const int W = c->seq.width - crop_left - crop_right, H = c->seq.height - crop_top - crop_bottom;
  if (crop_left < 0 || crop_right < 0 || crop_top < 0 || crop_bottom < 0 || W <= 0 || H <= 0) return HMGPU_EINVAL;
  hipSetDevice(c->device);
  Picture& p = c->pics[pic];
  size_t off[3], total = 0;
  for (int k = 0; k < 3; k++) { off[k] = total; total += align_up((size_t)(W >> (k ? c->csx : 0)) * bytes_per_sample * (H >> (k ? c->csy : 0)), 256); }
  uint8_t* d = static_cast<uint8_t*>(ctx_scratch(c, total));
  if (!d) return HMGPU_ENOMEM;
  hmgpu_status st = HMGPU_OK;
  for (int k = 0; k < 3 && st == HMGPU_OK; k++) {
    const int sx = k ? c->csx : 0, sy = k ? c->csy : 0, w = W >> sx, h = H >> sy;
    const int16_t* src = p.sao_applied ? p.dev.sao[k] : p.dev.rec[k];
    launch_pack(src, c->pitch[k], k ? kCStep : 1, crop_left >> sx, crop_top >> sy, w, h, bytes_per_sample, d + off[k], w * bytes_per_sample, c->stream);
    if (hipGetLastError() != hipSuccess ||
        hipMemcpy2DAsync(planes[k], (size_t)stride_bytes[k], d + off[k], (size_t)w * bytes_per_sample, (size_t)w * bytes_per_sample, h,
                         hipMemcpyDeviceToHost, c->stream) != hipSuccess) st = HMGPU_EDEVICE;
  }
  if (hipStreamSynchronize(c->stream) != hipSuccess) st = HMGPU_EDEVICE;
  prof_drain(c);
  const hmgpu_status fs = check_faults(c);
  return st != HMGPU_OK ? st : fs;
}

hmgpu_status hmgpu_picture_hash(hmgpu_ctx* c, hmgpu_pic pic, int32_t method, uint8_t digest[3][16], int32_t* digest_len) {
  if (!c || !valid_pic(c, pic) || !digest || !digest_len) return HMGPU_EINVAL;
  if (method == 1) {
    // MD5: one chain per plane (k_md5).  The call waits for it -- ~0.2 s for a 2160p luma plane; a decoder that must not wait uses
    // hmgpu_picture_hash_begin / hmgpu_hash_wait
    uint64_t t = 0;
    hmgpu_status st = hmgpu_picture_hash_begin(c, pic, 1, &t);
    int32_t ready = 0;
    if (st == HMGPU_OK) st = hmgpu_hash_wait(c, t, 1, digest, digest_len, &ready);
    return st;
  }
  if (method != 2 && method != 3) return HMGPU_EINVAL;
  hipSetDevice(c->device);
  Picture& p = c->pics[pic];
  const size_t words = 4 + (size_t)c->seq.height;          // three results + the per-row CRCs of one plane
  uint32_t* d = static_cast<uint32_t*>(ctx_scratch(c, words * 4));
  if (!d) return HMGPU_ENOMEM;
  hmgpu_status st = HMGPU_OK;
  if (hipMemsetAsync(d, 0, words * 4, c->stream) != hipSuccess) st = HMGPU_EDEVICE;
  for (int k = 0; k < 3 && st == HMGPU_OK; k++) {
    const int w = c->seq.width >> (k ? c->csx : 0), h = c->seq.height >> (k ? c->csy : 0);
    const int bd = k ? c->seq.bit_depth_chroma : c->seq.bit_depth_luma;
    const int16_t* src = p.sao_applied ? p.dev.sao[k] : p.dev.rec[k];
    if (method == 3) launch_checksum(src, c->pitch[k], k ? kCStep : 1, w, h, bd, d + k, c->stream);
    else launch_crc(src, c->pitch[k], k ? kCStep : 1, w, h, bd, d + 4, d + k, c->stream);
    if (hipGetLastError() != hipSuccess) st = HMGPU_EDEVICE;
  }
  uint32_t r[3] = {0, 0, 0};
  if (st == HMGPU_OK && hipMemcpyAsync(r, d, sizeof(r), hipMemcpyDeviceToHost, c->stream) != hipSuccess) st = HMGPU_EDEVICE;
  if (hipStreamSynchronize(c->stream) != hipSuccess) st = HMGPU_EDEVICE;
  if (st != HMGPU_OK) return st;
  memset(digest, 0, 48);
  for (int k = 0; k < 3; k++) {
    if (method == 2) { digest[k][0] = (uint8_t)(r[k] >> 8); digest[k][1] = (uint8_t)r[k]; }
    else { digest[k][0] = (uint8_t)(r[k] >> 24); digest[k][1] = (uint8_t)(r[k] >> 16); digest[k][2] = (uint8_t)(r[k] >> 8); digest[k][3] = (uint8_t)r[k]; }
  }
  *digest_len = method == 2 ? 2 : 4;
  return HMGPU_OK;
}

// the packed planes of a picture (HM's hash input, TComPicYuvMD5.cpp:44-84: rows of the visible area, 1 or 2 little-endian bytes per
// sample) + the digest words behind them
static size_t hash_slot_bytes(const hmgpu_ctx* c, size_t off[4]) {
  size_t total = 0;
  for (int k = 0; k < 3; k++) {
    const int bd = k ? c->seq.bit_depth_chroma : c->seq.bit_depth_luma;
    off[k] = total;
    total += align_up((size_t)(c->seq.width >> (k ? c->csx : 0)) * (c->seq.height >> (k ? c->csy : 0)) * (bd > 8 ? 2 : 1), 256);
  }
  off[3] = total;
  return total;
}

// the chains of the pictures handed over since the last launch: one lane per plane
static hmgpu_status hash_launch_pending(hmgpu_ctx* c) {
  if (c->hash_launched == c->hash_seq) return HMGPU_OK;
  size_t off[4];
  hash_slot_bytes(c, off);
  Md5Batch job;
  memset(&job, 0, sizeof(job));
  const int S = hmgpu_ctx::kHashSlots;
  int last_slot = 0;
  for (uint64_t t = c->hash_launched + 1; t <= c->hash_seq; t++) {
    const int slot = (int)(t % S);
    for (int k = 0; k < 3; k++) {
      const int nb = (k ? c->seq.bit_depth_chroma : c->seq.bit_depth_luma) > 8 ? 2 : 1;
      job.msg[job.n] = c->hash_buf[slot] + off[k];
      job.bytes[job.n] = (unsigned long long)(c->seq.width >> (k ? c->csx : 0)) * (c->seq.height >> (k ? c->csy : 0)) * nb;
      job.out[job.n] = c->hash_dev + slot * 12 + k * 4;
      job.n++;
    }
    last_slot = slot;
  }
  hipStream_t hs = c->hash_stream[c->hash_launches++ % hmgpu_ctx::kHashStreams];
  HIP_TRY(c, hipStreamWaitEvent(hs, c->hash_packed[last_slot], 0));       // (recorded in ticket order on the context's stream: the newest covers all)
  launch_md5(job, hs);
  HIP_TRY(c, hipGetLastError());
  for (uint64_t t = c->hash_launched + 1; t <= c->hash_seq; t++) {
    const int slot = (int)(t % S);
    HIP_TRY(c, hipMemcpyAsync(c->hash_host + slot * 12, c->hash_dev + slot * 12, 12 * sizeof(uint32_t), hipMemcpyDeviceToHost, hs));
    c->hash_done_slot[slot] = last_slot;
  }
  HIP_TRY(c, hipEventRecord(c->hash_done[last_slot], hs));
  c->hash_launched = c->hash_seq;
  return HMGPU_OK;
}

hmgpu_status hmgpu_picture_hash_begin(hmgpu_ctx* c, hmgpu_pic pic, int32_t method, uint64_t* ticket) {
  if (!c || !valid_pic(c, pic) || !ticket || method != 1) return HMGPU_EINVAL;
  hipSetDevice(c->device);
  const int S = hmgpu_ctx::kHashSlots;
  if (!c->hash_stream[0]) {
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    for (int k = 0; k < hmgpu_ctx::kHashStreams; k++) HIP_TRY(c, hipStreamCreateWithPriority(&c->hash_stream[k], hipStreamNonBlocking, lo));   // lowest priority: the chains fill gaps
    for (int k = 0; k < S; k++) {
      HIP_TRY(c, hipEventCreateWithFlags(&c->hash_packed[k], hipEventDisableTiming));
      HIP_TRY(c, hipEventCreateWithFlags(&c->hash_done[k], hipEventDisableTiming | hipEventBlockingSync));
    }
    HIP_TRY(c, hipMalloc((void**)&c->hash_dev, (size_t)S * 12 * sizeof(uint32_t)));
    HIP_TRY(c, hipHostMalloc((void**)&c->hash_host, (size_t)S * 12 * sizeof(uint32_t), hipHostMallocDefault));
  }
  const uint64_t t = c->hash_seq + 1;
  const int slot = (int)(t % S);
  size_t off[4];
  const size_t bytes = hash_slot_bytes(c, off);
  if (t > (uint64_t)S) {
    // the slot's previous chains (throttle: a caller that never waits is held back here once the ring is full)
    if (t - S > c->hash_launched) { const hmgpu_status st = hash_launch_pending(c); if (st != HMGPU_OK) return st; }
    HIP_TRY(c, hipEventSynchronize(c->hash_done[c->hash_done_slot[slot]]));
  }
  if (!c->hash_buf[slot]) HIP_TRY(c, hipMalloc((void**)&c->hash_buf[slot], bytes));
  Picture& p = c->pics[pic];
  uint8_t* d = c->hash_buf[slot];
  for (int k = 0; k < 3; k++) {
    const int w = c->seq.width >> (k ? c->csx : 0), h = c->seq.height >> (k ? c->csy : 0);
    const int nb = (k ? c->seq.bit_depth_chroma : c->seq.bit_depth_luma) > 8 ? 2 : 1;
    const int16_t* src = p.sao_applied ? p.dev.sao[k] : p.dev.rec[k];
    launch_pack(src, c->pitch[k], k ? kCStep : 1, 0, 0, w, h, nb, d + off[k], w * nb, c->stream);
  }
  HIP_TRY(c, hipGetLastError());
  // the picture itself is free again behind the packing; the chains run over the copy
  HIP_TRY(c, hipEventRecord(c->hash_packed[slot], c->stream));
  c->hash_seq = t;
  *ticket = t;
  touch(c, pic);
  commit_use(c);
  if (c->hash_seq - c->hash_launched >= (uint64_t)hmgpu_ctx::kHashBatch) return hash_launch_pending(c);
  return HMGPU_OK;
}

hmgpu_status hmgpu_hash_wait(hmgpu_ctx* c, uint64_t ticket, int32_t block, uint8_t digest[3][16], int32_t* digest_len, int32_t* ready) {
  if (!c || !digest || !digest_len || !ready || ticket == 0 || ticket > c->hash_seq || c->hash_seq - ticket >= (uint64_t)hmgpu_ctx::kHashSlots) return HMGPU_EINVAL;
  hipSetDevice(c->device);
  const int slot = (int)(ticket % hmgpu_ctx::kHashSlots);
  *ready = 0;
  if (ticket > c->hash_launched) {                          // its batch has not been launched yet: a waiting caller closes it
    if (!block) return HMGPU_OK;
    const hmgpu_status st = hash_launch_pending(c);
    if (st != HMGPU_OK) return st;
  }
  hipEvent_t ev = c->hash_done[c->hash_done_slot[slot]];
  if (block) { if (hipEventSynchronize(ev) != hipSuccess) return HMGPU_EDEVICE; }
  else {
    const hipError_t e = hipEventQuery(ev);
    if (e == hipErrorNotReady) { (void)hipGetLastError(); return HMGPU_OK; }
    if (e != hipSuccess) return HMGPU_EDEVICE;
  }
  const uint32_t* w = c->hash_host + slot * 12;
  for (int k = 0; k < 3; k++)
    for (int i = 0; i < 16; i++) digest[k][i] = (uint8_t)(w[k * 4 + (i >> 2)] >> (8 * (i & 3)));      // RFC 1321: the state words, low byte first
  *digest_len = 16;
  *ready = 1;
  return HMGPU_OK;
}

hmgpu_status hmgpu_picture_device_region(hmgpu_ctx* c, hmgpu_pic pic, int32_t which, void** base, int64_t* bytes) {
  if (!c || !valid_pic(c, pic) || !base || !bytes || (which != HMGPU_REGION_FINISHED && which != HMGPU_REGION_RECEIVE)) return HMGPU_EINVAL;
  hipSetDevice(c->device);
  Picture& p = c->pics[pic];
  const size_t plane_bytes = plane_set_bytes(c);
  if (which == HMGPU_REGION_FINISHED) {
    hmgpu_status st = ensure_extended(c, pic);
    if (st != HMGPU_OK) return st;
    *base = (char*)p.planes + (p.sao_applied ? plane_bytes : 0);     // rec planes first, SAO planes behind (alloc_picture)
  } else {
    if (p.sao_applied) {               // a received picture lives in the reconstruction planes, like an uploaded one
      p.sao_applied = false; p.dev.sao_applied = 0;
      hmgpu_status st = push_final(c, pic);
      if (st == HMGPU_OK) st = push_picdev(c, pic);
      if (st != HMGPU_OK) return st;
    }
    p.extended = false;
    *base = p.planes;
  }
  *bytes = (int64_t)plane_bytes;
  return HMGPU_OK;
}

hmgpu_status hmgpu_picture_commit_received(hmgpu_ctx* c, hmgpu_pic pic) {
  if (!c || !valid_pic(c, pic)) return HMGPU_EINVAL;
  Picture& p = c->pics[pic];
  if (p.sao_applied) return HMGPU_EINVAL;          // hmgpu_picture_device_region(RECEIVE) was not called
  p.extended = true;                               // the margins travelled with the planes
  return HMGPU_OK;
}

// A finished picture of one context into a picture of another context of the same geometry -- on another GPU of the process (a peer
// copy over xGMI) or on the same one -- ordered behind the work of both contexts' streams: what a decoder that spreads the pictures of
// one temporal level over several devices does with each reference picture (TComPrediction.cpp:593 reads it on the other device).
hmgpu_status hmgpu_picture_transfer(hmgpu_ctx* src, hmgpu_pic src_pic, hmgpu_ctx* dst, hmgpu_pic dst_pic) {
  if (!src || !dst || !valid_pic(src, src_pic) || !valid_pic(dst, dst_pic) || (src == dst && src_pic == dst_pic)) return HMGPU_EINVAL;
  const hmgpu_seq_params &a = src->seq, &b = dst->seq;
  if (a.width != b.width || a.height != b.height || a.log2_ctu_size != b.log2_ctu_size || a.bit_depth_luma != b.bit_depth_luma ||
      a.bit_depth_chroma != b.bit_depth_chroma) return HMGPU_EINVAL;
  void *from = nullptr, *to = nullptr;
  int64_t n_from = 0, n_to = 0;
  hmgpu_status st = hmgpu_picture_device_region(src, src_pic, HMGPU_REGION_FINISHED, &from, &n_from);      // (extends the border if needed)
  if (st == HMGPU_OK) st = hmgpu_picture_device_region(dst, dst_pic, HMGPU_REGION_RECEIVE, &to, &n_to);
  if (st != HMGPU_OK) return st;
  if (n_from != n_to) return HMGPU_EINVAL;
  if (src->device != dst->device) {
    int can = 0;
    if (hipDeviceCanAccessPeer(&can, dst->device, src->device) == hipSuccess && can) {
      hipSetDevice(dst->device);
      const hipError_t e = hipDeviceEnablePeerAccess(src->device, 0);
      if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return HMGPU_EDEVICE;
      (void)hipGetLastError();
    }
  }
  // source ready -> copy on the receiver's stream -> the source may be rewritten again
  hipSetDevice(src->device);
  if (!src->xfer_ev[0]) for (int k = 0; k < 2; k++) HIP_TRY(src, hipEventCreateWithFlags(&src->xfer_ev[k], hipEventDisableTiming));
  HIP_TRY(src, hipEventRecord(src->xfer_ev[0], src->stream));
  hipSetDevice(dst->device);
  HIP_TRY(dst, hipStreamWaitEvent(dst->stream, src->xfer_ev[0], 0));
  HIP_TRY(dst, hipMemcpyPeerAsync(to, dst->device, from, src->device, (size_t)n_from, dst->stream));
  HIP_TRY(dst, hipEventRecord(src->xfer_ev[1], dst->stream));
  hipSetDevice(src->device);
  HIP_TRY(src, hipStreamWaitEvent(src->stream, src->xfer_ev[1], 0));
  src->xfer_bytes += (uint64_t)n_from;
  touch(src, src_pic); commit_use(src);
  touch(dst, dst_pic); commit_use(dst);
  return hmgpu_picture_commit_received(dst, dst_pic);
}
uint64_t hmgpu_transfer_bytes(const hmgpu_ctx* c) { return c ? c->xfer_bytes : 0; }

void* hmgpu_stream(hmgpu_ctx* c) { return c ? (void*)c->stream : nullptr; }

// slice table entry of one slice (validation, SliceDev, scaling lists): the part of a slice call that does not depend on CTUs
static hmgpu_status register_slice(hmgpu_ctx* c, hmgpu_pic cur, int32_t slice_idx, const hmgpu_slice_params* sl, hipStream_t hs) {
  if (slice_idx < 0 || slice_idx >= HMGPU_MAX_SLICES || !sl) return HMGPU_EINVAL;
  if (sl->weighted_pred && (sl->wp_log2_denom[0] < 0 || sl->wp_log2_denom[0] > 7 || sl->wp_log2_denom[1] < 0 || sl->wp_log2_denom[1] > 7)) return HMGPU_EINVAL;
  Picture& p = c->pics[cur];
  // reference pictures must be live device pictures
  for (int l = 0; l < 2; l++) {
    if (sl->num_ref_idx[l] < 0 || sl->num_ref_idx[l] > HMGPU_MAX_REF) return HMGPU_EINVAL;
    for (int i = 0; i < sl->num_ref_idx[l]; i++) if (!valid_pic(c, sl->ref_pic[l][i]) || sl->ref_pic[l][i] == cur) return HMGPU_EINVAL;
  }
  hipSetDevice(c->device);
  SliceDev sd;
  memset(&sd, 0, sizeof(sd));
  sd.slice_type = sl->slice_type; sd.cb_qp_offset = sl->cb_qp_offset; sd.cr_qp_offset = sl->cr_qp_offset;
  sd.pps_cb_qp_offset = sl->pps_cb_qp_offset; sd.pps_cr_qp_offset = sl->pps_cr_qp_offset;
  sd.deblocking_disable = sl->deblocking_disable; sd.beta_offset_div2 = sl->beta_offset_div2; sd.tc_offset_div2 = sl->tc_offset_div2;
  sd.lf_across_slices = sl->lf_across_slices;
  sd.constrained_intra_pred = sl->constrained_intra_pred ? 1 : 0;
  sd.weighted_pred = sl->weighted_pred ? 1 : 0;
  sd.wp_log2_denom[0] = sl->wp_log2_denom[0]; sd.wp_log2_denom[1] = sl->wp_log2_denom[1];
  memcpy(sd.wp_weight, sl->wp_weight, sizeof(sd.wp_weight));
  memcpy(sd.wp_offset, sl->wp_offset, sizeof(sd.wp_offset));
  for (int l = 0; l < 2; l++)
    for (int i = 0; i < HMGPU_MAX_REF; i++) {
      sd.ref_poc[l][i] = i < sl->num_ref_idx[l] ? sl->ref_poc[l][i] : 0;
      sd.ref_pic[l][i] = i < sl->num_ref_idx[l] ? (int8_t)sl->ref_pic[l][i] : (int8_t)-1;
    }
  p.slices[slice_idx] = sd;
  p.max_slice = std::max(p.max_slice, (int)slice_idx);
  p.dev.lf_across_tiles = sl->lf_across_tiles;
  p.dev.sl_m = nullptr;
  if (sl->scaling_lists) {
    // xSetScalingListDec / processScalingListDec (TComTrQuant.cpp:2992-3012, 3092-3106) without the per-QP factor: m per position
    const hmgpu_scaling_lists& L = *sl->scaling_lists;
    p.sl_host.assign(4 * 6 * 1024, 16);
    for (int sz = 0; sz < 4; sz++)
      for (int l = 0; l < 6; l++) {
        const int n = 4 << sz, ratio = n > 8 ? n / 8 : 1, mn = n > 8 ? 8 : n;
        uint8_t* t = p.sl_host.data() + (sz * 6 + l) * 1024;
        for (int y = 0; y < n; y++)
          for (int x = 0; x < n; x++) {
            const int v = (ratio > 1 && x == 0 && y == 0) ? L.dc[sz][l] : L.coef[sz][l][mn * (y / ratio) + x / ratio];
            if (v < 1 || v > 255) return HMGPU_EINVAL;
            t[y * n + x] = (uint8_t)v;
          }
      }
    HIP_TRY(c, h2d_small(c, p.sl_table, p.sl_host.data(), p.sl_host.size(), hs));
    p.dev.sl_m = p.sl_table;
  }
  HIP_TRY(c, h2d_small(c, (void*)(p.dev.slices + slice_idx), &p.slices[slice_idx], sizeof(SliceDev), hs));
  return HMGPU_OK;
}

static bool stg_starts_contiguous(const hmgpu_coeffs* co, int num_ctus) {
  return co->ctu_level_start[1] == co->ctu_level_start[0] + (num_ctus + 1) && co->ctu_level_start[2] == co->ctu_level_start[1] + (num_ctus + 1);
}

// a staging block whose arrays the caller handed over for a whole picture: its metadata is ONE copy, its levels another
static const hmgpu_staging* staging_of(const hmgpu_ctx* c, const hmgpu_ctu_meta* m, const hmgpu_coeffs* co) {
  for (size_t i = 0; i < c->stagings.size() + c->shared_stagings.size(); i++) {
    const hmgpu_staging* st = i < c->stagings.size() ? c->stagings[i] : c->shared_stagings[i - c->stagings.size()];
    const hmgpu_ctu_meta& h = st->m;
    if (m->depth != h.depth) continue;
    // the required arrays are the block's; the optional ones are the block's or left out (NULL: that group does not travel)
    bool ok = m->part_size == h.part_size && m->pred_mode == h.pred_mode && m->qp == h.qp && m->tr_idx == h.tr_idx && m->slice_idx == h.slice_idx &&
              m->tile_idx == h.tile_idx;
    for (int k = 0; k < 3 && ok; k++) ok = m->cbf[k] == h.cbf[k] && (!m->transform_skip[k] || m->transform_skip[k] == h.transform_skip[k]);
    for (int k = 0; k < 2 && ok; k++) ok = m->mv[k] == h.mv[k] && m->ref_idx[k] == h.ref_idx[k] && (!m->intra_dir[k] || m->intra_dir[k] == h.intra_dir[k]);
    ok = ok && (!m->transquant_bypass || m->transquant_bypass == h.transquant_bypass) && (!m->ipcm || m->ipcm == h.ipcm);
    for (int k = 0; k < 3 && ok; k++) ok = co->level[k] == st->co.level[k];
    if (ok) return st;
  }
  return nullptr;
}

// HM arrays of a CTU range to the device (on stream hs) and the record of the call.  `slices` lists the slice table entries whose
// reference pictures the range may read; slice_idx is the one a missing meta->slice_idx array stands for.
static hmgpu_status stage_inputs(hmgpu_ctx* c, hmgpu_pic cur, int32_t slice_idx, const std::vector<int>& slices, bool any_wp,
                                 const hmgpu_ctu_meta* m, const hmgpu_coeffs* co, int32_t first_ctu, int32_t num_ctus, hipStream_t hs,
                                 SliceCall* call_out) {
  Picture& p = c->pics[cur];
  const size_t po = (size_t)first_ctu * c->parts, pn = (size_t)num_ctus * c->parts;
  // lossless / PCM CUs need their own inputs
  const bool any_pcm = m->ipcm && memchr(m->ipcm + po, 1, pn) != nullptr;
  const bool any_bypass = m->transquant_bypass && memchr(m->transquant_bypass + po, 1, pn) != nullptr;
  if (any_pcm && (!co->pcm_sample[0] || !co->pcm_sample[1] || !co->pcm_sample[2] || !m->intra_dir[0])) return HMGPU_EINVAL;
  if (any_pcm && (c->seq.pcm_bit_depth_luma < 1 || c->seq.pcm_bit_depth_luma > c->seq.bit_depth_luma ||
                  c->seq.pcm_bit_depth_chroma < 1 || c->seq.pcm_bit_depth_chroma > c->seq.bit_depth_chroma)) return HMGPU_EINVAL;
  p.dev.has_intra_dir = (m->intra_dir[0] && m->intra_dir[1]) ? 1 : 0;      // without the modes intra CUs are left untouched
  const hmgpu_staging* stg = (first_ctu == 0 && num_ctus == c->num_ctus) ? staging_of(c, m, co) : nullptr;
  const bool compact = co->ctu_level_start[0] && co->ctu_level_start[1] && co->ctu_level_start[2];
  if (!compact && (co->ctu_level_start[0] || co->ctu_level_start[1] || co->ctu_level_start[2])) return HMGPU_EINVAL;
  if (compact) {
    if (c->fmt != 1) return HMGPU_EUNSUPPORTED;                              // (4:2:2 / 4:4:4: HM's dense layout only)
    if (first_ctu != 0 || num_ctus != c->num_ctus) return HMGPU_EINVAL;      // whole pictures only
    for (int k = 0; k < 3; k++) {
      // the CTUs' pieces follow each other and none is longer than a CTU (k_intra stages a CTU's piece into LDS by these numbers)
      const uint32_t per = (uint32_t)(c->ctu * c->ctu) >> (k ? 2 : 0);
      const uint32_t* st = co->ctu_level_start[k];
      if (st[c->num_ctus] > c->coef_elems[k]) return HMGPU_EINVAL;
      for (int i = 0; i < c->num_ctus; i++) if (st[i + 1] < st[i] || st[i + 1] - st[i] > per) return HMGPU_EINVAL;
    }
    // the CTU starts (from a staging block: its three arrays in one copy)
    const bool one = stg_starts_contiguous(co, c->num_ctus);
    for (int k = 0; k < (one ? 1 : 3); k++)
      HIP_TRY(c, hipMemcpyAsync(p.coef_start + (size_t)k * (c->num_ctus + 1), co->ctu_level_start[k],
                                (size_t)(one ? 3 : 1) * (c->num_ctus + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, hs));
    for (int k = 0; k < 3; k++) {
      p.dev.coef_start[k] = p.coef_start + (size_t)k * (c->num_ctus + 1);
      const size_t n = co->ctu_level_start[k][c->num_ctus];
      if (n) HIP_TRY(c, hipMemcpyAsync((void*)p.dev.coef[k], co->level[k], n * sizeof(int16_t), hipMemcpyHostToDevice, hs));
    }
  } else {
    for (int k = 0; k < 3; k++) p.dev.coef_start[k] = nullptr;
  }
  if (stg) {
    // the caller filled a staging block: the metadata block in one DMA (the dense levels in another) -- minus the groups this
    // picture does without: list 1 when no slice is a B slice (k_prep ignores it then), the intra modes and the transform-skip /
    // lossless / PCM flags when the caller left them out (the device copies of the flags are cleared if an earlier picture set them)
    bool any_b_slice = false;
    for (int si : slices) any_b_slice |= p.slices[si].slice_type == HMGPU_B_SLICE;
    const bool flags_used = m->transform_skip[0] || m->transform_skip[1] || m->transform_skip[2] || m->transquant_bypass || m->ipcm;
    const bool want[4] = {true, any_b_slice, p.dev.has_intra_dir != 0, flags_used};
    for (int g0 = 0; g0 < 4;) {
      if (!want[g0]) { g0++; continue; }
      int g1 = g0 + 1;
      while (g1 < 4 && want[g1]) g1++;
      HIP_TRY(c, hipMemcpyAsync((char*)p.meta + stg->grp[g0], stg->host + stg->grp[g0], stg->grp[g1] - stg->grp[g0], hipMemcpyHostToDevice, hs));
      g0 = g1;
    }
    if (!flags_used && p.flags_staged) HIP_TRY(c, hipMemsetAsync((char*)p.meta + stg->grp[3], 0, stg->grp[4] - stg->grp[3], hs));
    p.flags_staged = flags_used;
    if (!compact) HIP_TRY(c, hipMemcpyAsync(p.coef, stg->host + stg->meta_bytes, stg->coef_bytes, hipMemcpyHostToDevice, hs));
    p.h_slice_idx.assign(m->slice_idx, m->slice_idx + c->num_ctus);
    p.h_tile_idx.assign(m->tile_idx, m->tile_idx + c->num_ctus);
  } else {
    ProfScope ps(c, K_H2D);
    p.flags_staged = true;
    // ---- HM arrays of the CTU range (field-by-field, exactly the arrays TComDataCU owns)
#define STAGE(dst, src, elem_bytes)                                                                                       \
    if (src) HIP_TRY(c, hipMemcpyAsync((char*)(dst) + po * (elem_bytes), (const char*)(src) + po * (elem_bytes), pn * (elem_bytes), \
                                       hipMemcpyHostToDevice, hs));                                                \
    else HIP_TRY(c, hipMemsetAsync((char*)(dst) + po * (elem_bytes), 0, pn * (elem_bytes), hs))
    STAGE(p.dev.depth, m->depth, 1); STAGE(p.dev.part_size, m->part_size, 1); STAGE(p.dev.pred_mode, m->pred_mode, 1);
    STAGE(p.dev.qp, m->qp, 1); STAGE(p.dev.tr_idx, m->tr_idx, 1);
    for (int k = 0; k < 3; k++) { STAGE(p.dev.cbf[k], m->cbf[k], 1); STAGE(p.dev.tskip[k], m->transform_skip[k], 1); }
    for (int k = 0; k < 2; k++) { STAGE(p.dev.mv[k], m->mv[k], 4); STAGE(p.dev.ref_idx[k], m->ref_idx[k], 1); }
    if (p.dev.has_intra_dir) { STAGE(p.dev.intra_dir[0], m->intra_dir[0], 1); STAGE(p.dev.intra_dir[1], m->intra_dir[1], 1); }
    STAGE(p.dev.bypass, m->transquant_bypass, 1); STAGE(p.dev.ipcm, m->ipcm, 1);
#undef STAGE
    // per-CTU slice / tile index (the slice index of this call wins over a missing array)
    {
      // (the host mirrors are what the asynchronous copies read from: they live as long as the picture)
      p.h_slice_idx.resize(c->num_ctus);
      p.h_tile_idx.resize(c->num_ctus);
      for (int i = 0; i < num_ctus; i++) p.h_slice_idx[first_ctu + i] = m->slice_idx ? m->slice_idx[first_ctu + i] : (uint16_t)slice_idx;
      for (int i = 0; i < num_ctus; i++) p.h_tile_idx[first_ctu + i] = m->tile_idx ? m->tile_idx[first_ctu + i] : (uint16_t)0;
      HIP_TRY(c, h2d_small(c, (void*)(p.dev.slice_idx + first_ctu), p.h_slice_idx.data() + first_ctu, (size_t)num_ctus * 2, hs));
      HIP_TRY(c, h2d_small(c, (void*)(p.dev.tile_idx + first_ctu), p.h_tile_idx.data() + first_ctu, (size_t)num_ctus * 2, hs));
    }
    for (int k = 0; k < 3 && !compact; k++) {
      const size_t per = (size_t)(c->ctu * c->ctu) >> (k ? c->csx + c->csy : 0);
      HIP_TRY(c, hipMemcpyAsync((void*)(p.dev.coef[k] + first_ctu * per), co->level[k] + first_ctu * per, (size_t)num_ctus * per * 2,
                                hipMemcpyHostToDevice, hs));
    }
  }
  // cross-component prediction weights (4:4:4; m_crossComponentPredictionAlpha): device copies allocated with the first picture that carries them
  p.dev.ccp[0] = p.dev.ccp[1] = nullptr;
  if (c->fmt == 3 && m->ccp_alpha[0] && m->ccp_alpha[1]) {
    const size_t np = (size_t)c->num_ctus * c->parts;
    if (!p.ccp) HIP_TRY(c, hipMalloc(&p.ccp, 2 * np));
    for (int k = 0; k < 2; k++) {
      HIP_TRY(c, hipMemcpyAsync((char*)p.ccp + k * np + po, m->ccp_alpha[k] + po, pn, hipMemcpyHostToDevice, hs));
      p.dev.ccp[k] = (const int8_t*)p.ccp + k * np;
    }
  }
  {
    if (any_pcm) {
      size_t bytes = 0;
      for (int k = 0; k < 3; k++) bytes += align_up(c->coef_elems[k] * sizeof(int16_t), 256);
      if (!p.pcm) {
        HIP_TRY(c, hipMalloc(&p.pcm, bytes));
        Carver cp(p.pcm);
        for (int k = 0; k < 3; k++) p.dev.pcm[k] = cp.take<int16_t>(c->coef_elems[k]);
      }
      for (int k = 0; k < 3; k++) {
        const size_t per = (size_t)(c->ctu * c->ctu) >> (k ? c->csx + c->csy : 0);
        HIP_TRY(c, hipMemcpyAsync((void*)(p.dev.pcm[k] + first_ctu * per), co->pcm_sample[k] + first_ctu * per, (size_t)num_ctus * per * 2,
                                  hipMemcpyHostToDevice, hs));
      }
      p.dev.pcm_shift[0] = c->seq.bit_depth_luma - c->seq.pcm_bit_depth_luma;
      p.dev.pcm_shift[1] = p.dev.pcm_shift[2] = c->seq.bit_depth_chroma - c->seq.pcm_bit_depth_chroma;
    }
    if (any_bypass || (any_pcm && c->seq.pcm_loop_filter_disable)) p.dev.any_nofilt = 1;
    HIP_TRY(c, h2d_small(c, c->d_pics + cur, &p.dev, sizeof(PicDev), hs));
  }
  // a range decoded again (picture buffer reused without release/acquire) replaces the earlier record
  p.calls.erase(std::remove_if(p.calls.begin(), p.calls.end(), [&](const SliceCall& o) {
                  return o.first_ctu < first_ctu + num_ctus && first_ctu < o.first_ctu + o.num_ctus; }), p.calls.end());
  // The caller's arrays are at hand: ONE pass over three of them (branch-free, so that the compiler vectorises it: ~1.5 MB per 2160p picture)
  // says whether the range holds intra CUs at all and how many (which intra kernel, if any: launch_intra) and whether it holds PUs that cut
  // an 8x8 luma tile -- 2NxN / Nx2N (/ NxN) parts of 8x8 CUs, the 4- and 12-sample parts of AMP in 16x16 CUs -- (the cells kernels).
  // (Round 4: the search for such PUs was a loop with an early exit over every 8x8 area; on pictures without them it walked all of them,
  // 0.2 ms of the calling thread per 2160p picture; this pass takes ~0.05.)
  size_t n_intra = 0;
  unsigned cells_u = 0;
  {
    // (byte lanes throughout -- 16 or 32 partitions per vector instruction --: the counts of a chunk of 192 stay below 256)
    const uint8_t d8 = (uint8_t)(c->seq.log2_ctu_size - 3), d8m = (uint8_t)(d8 - 1);
    const int8_t* __restrict ps = m->part_size + po;
    const uint8_t* __restrict dp = m->depth + po;
    const int8_t* __restrict pm = m->pred_mode + po;
    for (size_t base = 0; base < pn; base += 192) {
      const size_t n = std::min<size_t>(192, pn - base);
      uint8_t cnt = 0, cel = 0;
      for (size_t i = 0; i < n; i++) {
        const uint8_t ptn = (uint8_t)ps[base + i], d = dp[base + i];
        const uint8_t intra = (uint8_t)(pm[base + i] == HMGPU_MODE_INTRA);
        const uint8_t part = (uint8_t)((ptn != HMGPU_SIZE_2Nx2N) & (ptn != HMGPU_SIZE_NONE));
        const uint8_t small = (uint8_t)((d >= d8) | ((d == d8m) & (ptn >= HMGPU_SIZE_2NxnU)));
        cnt = (uint8_t)(cnt + intra);
        cel = (uint8_t)(cel | (part & (intra ^ 1) & small));
      }
      n_intra += cnt; cells_u |= cel;
    }
  }
  const bool has_intra = p.dev.has_intra_dir && n_intra != 0;
  const bool cells = cells_u != 0;
  bool any_b = false, any_i = false;
  for (int si : slices) { any_b |= p.slices[si].slice_type == HMGPU_B_SLICE; any_i |= p.slices[si].slice_type == HMGPU_I_SLICE; }
  // I slices, or a range at least half intra, take the intra kernel that stages whole CTUs
  if (has_intra && !any_i) any_i = 2 * n_intra >= pn;
  SliceCall call = {first_ctu, num_ctus, slice_idx, has_intra, any_wp, cells, any_b, any_i};
  p.calls.push_back(call);
  p.extended = false;
  *call_out = call;
  return HMGPU_OK;
}

// reference pictures named by the slice table entries `slices` of picture `cur`: their borders must be extended before the kernels read them
static hmgpu_status extend_refs_of(hmgpu_ctx* c, hmgpu_pic cur, const std::vector<int>& slices) {
  const Picture& p = c->pics[cur];
  for (int si : slices) {
    const SliceDev& sd = p.slices[si];
    for (int l = 0; l < 2; l++)
      for (int r = 0; r < HMGPU_MAX_REF; r++)
        if (sd.ref_pic[l][r] >= 0) { hmgpu_status st = ensure_extended(c, sd.ref_pic[l][r]); if (st != HMGPU_OK) return st; }
  }
  return HMGPU_OK;
}

// staging + the reconstruction kernels of ONE call, everything on the context's stream
static hmgpu_status stage_and_run(hmgpu_ctx* c, hmgpu_pic cur, int32_t slice_idx, const std::vector<int>& slices, bool any_wp,
                                  const hmgpu_ctu_meta* m, const hmgpu_coeffs* co, int32_t first_ctu, int32_t num_ctus) {
  SliceCall call;
  hmgpu_status st = stage_inputs(c, cur, slice_idx, slices, any_wp, m, co, first_ctu, num_ctus, c->stream, &call);
  if (st == HMGPU_OK) st = extend_refs_of(c, cur, slices);
  if (st != HMGPU_OK) return st;
  Batch b; memset(&b, 0, sizeof(b));
  b.n = 1; b.pic[0] = cur; b.first_ctu[0] = first_ctu; b.num_ctus[0] = num_ctus;
  st = run_recon(c, b, call.intra, call.wp, call.cells, call.bi, call.islice);
  mark_use(c, b);
  return st;
}

static bool meta_complete(const hmgpu_ctu_meta* m, const hmgpu_coeffs* co) {
  return m && co && m->depth && m->part_size && m->pred_mode && m->qp && m->tr_idx && m->cbf[0] && m->cbf[1] && m->cbf[2] && m->mv[0] &&
         m->mv[1] && m->ref_idx[0] && m->ref_idx[1] && co->level[0] && co->level[1] && co->level[2];
}

static hmgpu_status reopen_picture(hmgpu_ctx* c, hmgpu_pic cur) {
  Picture& p = c->pics[cur];
  if (p.sao_applied) {                 // picture buffer decoded again without release/acquire: reconstruction planes again
    p.sao_applied = false; p.dev.sao_applied = 0;
    return push_final(c, cur);
  }
  return HMGPU_OK;
}

hmgpu_status hmgpu_decompress_slice(hmgpu_ctx* c, hmgpu_pic cur, int32_t slice_idx, const hmgpu_slice_params* sl,
                                    const hmgpu_ctu_meta* m, const hmgpu_coeffs* co, int32_t first_ctu, int32_t num_ctus) {
  if (!c || !valid_pic(c, cur) || !sl || !meta_complete(m, co)) return HMGPU_EINVAL;
  if (first_ctu < 0 || num_ctus <= 0 || first_ctu + num_ctus > c->num_ctus) return HMGPU_EINVAL;
  hmgpu_status st = reopen_picture(c, cur);
  if (st == HMGPU_OK) st = register_slice(c, cur, slice_idx, sl, c->stream);
  if (st != HMGPU_OK) return st;
  return stage_and_run(c, cur, slice_idx, std::vector<int>{slice_idx}, sl->weighted_pred != 0, m, co, first_ctu, num_ctus);
}

hmgpu_status hmgpu_decompress_picture(hmgpu_ctx* c, hmgpu_pic cur, int32_t num_slices, const hmgpu_slice_params* const* slices,
                                      const hmgpu_ctu_meta* m, const hmgpu_coeffs* co) {
  if (!c || !valid_pic(c, cur) || !slices || num_slices < 1 || num_slices > HMGPU_MAX_SLICES || !meta_complete(m, co)) return HMGPU_EINVAL;
  if (num_slices > 1 && !m->slice_idx) return HMGPU_EINVAL;
  if (m->slice_idx) for (int i = 0; i < c->num_ctus; i++) if (m->slice_idx[i] >= num_slices) return HMGPU_EINVAL;
  hmgpu_status st = reopen_picture(c, cur);
  std::vector<int> all;
  bool any_wp = false;
  for (int i = 0; i < num_slices && st == HMGPU_OK; i++) {
    st = register_slice(c, cur, i, slices[i], c->stream);
    all.push_back(i);
    any_wp |= slices[i] && slices[i]->weighted_pred != 0;
  }
  if (st != HMGPU_OK) return st;
  return stage_and_run(c, cur, 0, all, any_wp, m, co, 0, c->num_ctus);
}

// HM's dense level arrays -> compact streams.  The same walk as k_prep's count (k_prep.hip): per 8x8 luma area in z-order, the TUs
// that originate there; a TU is coded iff its cbf bits are set down to its transform depth.
hmgpu_status hmgpu_pack_levels(const hmgpu_seq_params* seq, const hmgpu_ctu_meta* m, const hmgpu_coeffs* dense,
                               int16_t* const out_level[3], uint32_t* const out_start[3]) {
  if (!seq || !m || !dense || !out_level || !out_start || !m->depth || !m->part_size || !m->tr_idx || !m->cbf[0] || !m->cbf[1] || !m->cbf[2]) return HMGPU_EINVAL;
  for (int k = 0; k < 3; k++) if (!dense->level[k] || !out_level[k] || !out_start[k]) return HMGPU_EINVAL;
  if (seq->chroma_format > 1) return HMGPU_EUNSUPPORTED;       // (the compact form is defined for 4:2:0 / 4:0:0 pictures)
  const int log2ctu = seq->log2_ctu_size, ctu_sz = 1 << log2ctu, pw = ctu_sz / 4, parts = pw * pw;
  const int ctus_w = (seq->width + ctu_sz - 1) / ctu_sz, n_ctus = hmgpu_num_ctus(seq);
  uint32_t pos[3] = {0, 0, 0};
  for (int a = 0; a < n_ctus; a++) {
    const int cx = (a % ctus_w) * ctu_sz, cy = (a / ctus_w) * ctu_sz;
    for (int k = 0; k < 3; k++) out_start[k][a] = pos[k];
    for (int z0 = 0; z0 < parts; z0 += 4) {
      const size_t idx = (size_t)a * parts + z0;
      const int x4 = zscan_x(z0), y4 = zscan_y(z0);
      if (cx + 4 * x4 >= seq->width || cy + 4 * y4 >= seq->height || m->part_size[idx] == HMGPU_SIZE_NONE) continue;
      const int tr = m->tr_idx[idx], log2tu = log2ctu - m->depth[idx] - tr;
      if (log2tu > 5) continue;
      const unsigned chain = (1u << (tr + 1)) - 1;
      auto emit = [&](int comp, size_t src_off, uint32_t n) {
        memcpy(out_level[comp] + pos[comp], dense->level[comp] + src_off, n * sizeof(int16_t));
        pos[comp] += n;
      };
      const size_t base_l = (size_t)a * ctu_sz * ctu_sz, base_c = base_l / 4;
      if (log2tu > 2) {
        const int tu_parts = 1 << (log2tu - 2);
        if ((x4 & (tu_parts - 1)) || (y4 & (tu_parts - 1))) continue;
        if ((m->cbf[0][idx] & chain) == chain) emit(0, base_l + 16 * (size_t)z0, 1u << (2 * log2tu));
        if ((m->cbf[1][idx] & chain) == chain) emit(1, base_c + 4 * (size_t)z0, 1u << (2 * log2tu - 2));
        if ((m->cbf[2][idx] & chain) == chain) emit(2, base_c + 4 * (size_t)z0, 1u << (2 * log2tu - 2));
      } else {
        for (int j = 0; j < 4; j++) if ((m->cbf[0][idx + j] & chain) == chain) emit(0, base_l + 16 * (size_t)(z0 + j), 16);
        if ((m->cbf[1][idx] & chain) == chain) emit(1, base_c + 4 * (size_t)z0, 16);
        if ((m->cbf[2][idx] & chain) == chain) emit(2, base_c + 4 * (size_t)z0, 16);
      }
    }
  }
  for (int k = 0; k < 3; k++) out_start[k][n_ctus] = pos[k];
  return HMGPU_OK;
}

// ---- staging blocks
hmgpu_status hmgpu_staging_alloc(hmgpu_ctx* c, hmgpu_staging** out, hmgpu_ctu_meta* meta, hmgpu_coeffs* coeffs) {
  if (!c || !out || !meta || !coeffs) return HMGPU_EINVAL;
  hipSetDevice(c->device);
  hmgpu_staging* st = new (std::nothrow) hmgpu_staging();
  if (!st) return HMGPU_ENOMEM;
  const size_t np = (size_t)c->num_ctus * c->parts;
  PicDev lay;
  memset(&lay, 0, sizeof(lay));
  { Carver m(nullptr); carve_meta(m, lay, np, c->num_ctus, st->grp); st->meta_bytes = m.off; }
  for (int k = 0; k < 3; k++) st->coef_bytes += align_up(c->coef_elems[k] * sizeof(int16_t), 256);
  st->start_bytes = align_up((size_t)3 * (c->num_ctus + 1) * sizeof(uint32_t), 256);
  const size_t total = st->meta_bytes + st->coef_bytes + st->start_bytes;
  // (portable: every device of the process may copy from the block -- hmgpu_staging_share)
  if (hipHostMalloc((void**)&st->host, total, hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); delete st; return HMGPU_ENOMEM; }
  st->owner = c;
  memset(st->host, 0, total);
  { Carver m(st->host); carve_meta(m, lay, np, c->num_ctus); }
  hmgpu_ctu_meta& h = st->m;
  memset(&h, 0, sizeof(h));
  h.depth = lay.depth; h.part_size = lay.part_size; h.pred_mode = lay.pred_mode; h.qp = lay.qp; h.tr_idx = lay.tr_idx;
  for (int k = 0; k < 3; k++) { h.cbf[k] = lay.cbf[k]; h.transform_skip[k] = lay.tskip[k]; }
  for (int k = 0; k < 2; k++) { h.mv[k] = lay.mv[k]; h.ref_idx[k] = lay.ref_idx[k]; h.intra_dir[k] = lay.intra_dir[k]; }
  h.transquant_bypass = lay.bypass; h.ipcm = lay.ipcm; h.slice_idx = lay.slice_idx; h.tile_idx = lay.tile_idx;
  // (decoded nowhere yet: HM marks that with part_size = NUMBER_OF_PART_SIZES and ref_idx = -1)
  memset(const_cast<int8_t*>(h.part_size), HMGPU_SIZE_NONE, np);
  memset(const_cast<int8_t*>(h.ref_idx[0]), 0xff, np); memset(const_cast<int8_t*>(h.ref_idx[1]), 0xff, np);
  memset(&st->co, 0, sizeof(st->co));
  { Carver m(st->host + st->meta_bytes); for (int k = 0; k < 3; k++) st->co.level[k] = m.take<int16_t>(c->coef_elems[k]); }
  for (int k = 0; k < 3; k++) st->co.ctu_level_start[k] = reinterpret_cast<const uint32_t*>(st->host + st->meta_bytes + st->coef_bytes) + (size_t)k * (c->num_ctus + 1);
  c->stagings.push_back(st);
  *meta = st->m; *coeffs = st->co; *out = st;
  return HMGPU_OK;
}

// the block may be rewritten once the copies of the call that last read it have been made (events of the copy stream are recorded in
// order: one that has since been re-recorded stands for a later point of the same stream)
hmgpu_status hmgpu_staging_wait(hmgpu_ctx* c, hmgpu_staging* st) {
  if (!c || !st) return HMGPU_EINVAL;
  if (st->copy_seq == 0) return HMGPU_OK;
  hmgpu_ctx* r = st->reader ? st->reader : c;              // the context whose copy stream read the block last
  hipSetDevice(r->device);
  if (hipEventSynchronize(r->copy_ev[st->copy_seq % 8]) != hipSuccess) return HMGPU_EDEVICE;
  return HMGPU_OK;
}

// A decoder that places pictures on several contexts parses into ONE set of blocks and decides late which context decodes a picture:
// `other` -- a context of the same geometry, on any device -- recognises the block's arrays from now on as `owner` does.
hmgpu_status hmgpu_staging_share(hmgpu_ctx* owner, hmgpu_staging* st, hmgpu_ctx* other) {
  if (!owner || !st || !other || st->owner != owner) return HMGPU_EINVAL;
  if (other == owner || std::find(st->sharers.begin(), st->sharers.end(), other) != st->sharers.end()) return HMGPU_OK;
  const hmgpu_seq_params &a = owner->seq, &b = other->seq;
  if (a.width != b.width || a.height != b.height || a.log2_ctu_size != b.log2_ctu_size || a.chroma_format != b.chroma_format ||
      owner->num_ctus != other->num_ctus || owner->parts != other->parts) return HMGPU_EINVAL;
  for (int k = 0; k < 3; k++) if (owner->coef_elems[k] != other->coef_elems[k]) return HMGPU_EINVAL;
  st->sharers.push_back(other);
  other->shared_stagings.push_back(st);
  return HMGPU_OK;
}

void hmgpu_staging_free(hmgpu_ctx* c, hmgpu_staging* st) {
  if (!c || !st) return;
  hipSetDevice(c->device);
  if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
  if (st->reader && st->reader != c && st->reader->copy_stream) { hipSetDevice(st->reader->device); (void)hipStreamSynchronize(st->reader->copy_stream); }
  for (hmgpu_ctx* o : st->sharers) o->shared_stagings.erase(std::remove(o->shared_stagings.begin(), o->shared_stagings.end(), st), o->shared_stagings.end());
  c->stagings.erase(std::remove(c->stagings.begin(), c->stagings.end(), st), c->stagings.end());
  if (st->host) (void)hipHostFree(st->host);
  delete st;
}

// the copy stream may overwrite a picture's input arrays once the kernels that last read them have finished
static void wait_for_last_use(hmgpu_ctx* c, const Picture& p, hipStream_t hs) {
  if (!p.last_use) return;
  // (events older than the ring are gone: the newest one was recorded later and is a safe stand-in)
  const uint64_t seq = c->use_seq - p.last_use < 8 ? p.last_use : c->use_seq;
  (void)hipStreamWaitEvent(hs, c->use_ev[seq % 8], 0);
}

hmgpu_status hmgpu_decompress_pictures(hmgpu_ctx* c, int32_t n, const hmgpu_picture_job* jobs) {
  if (!c || !jobs || n < 1 || n > kMaxBatch) return HMGPU_EINVAL;
  c->host_calls++;
  { HostTimer tv(c, 0);
  for (int i = 0; i < n; i++) {
    const hmgpu_picture_job& j = jobs[i];
    if (!valid_pic(c, j.pic) || !j.slices || j.num_slices < 1 || j.num_slices > HMGPU_MAX_SLICES || !meta_complete(j.meta, j.coeffs)) return HMGPU_EINVAL;
    if (j.num_slices > 1 && !j.meta->slice_idx) return HMGPU_EINVAL;
    if (j.meta->slice_idx) for (int k = 0; k < c->num_ctus; k++) if (j.meta->slice_idx[k] >= j.num_slices) return HMGPU_EINVAL;
    for (int k = 0; k < i; k++) if (jobs[k].pic == j.pic) return HMGPU_EINVAL;
    // independent pictures only: none of them may be a reference of another one of the call
    for (int s2 = 0; s2 < j.num_slices; s2++)
      for (int l = 0; l < 2 && j.slices[s2]; l++)
        for (int r = 0; r < j.slices[s2]->num_ref_idx[l] && r < HMGPU_MAX_REF; r++)
          for (int k = 0; k < n; k++) if (j.slices[s2]->ref_pic[l][r] == jobs[k].pic) return HMGPU_EINVAL;
  }
  }
  hipSetDevice(c->device);
  Batch b; memset(&b, 0, sizeof(b));
  b.n = n;
  bool any_intra = false, any_wp = false, any_cells = false, any_bi = false, any_islice = false;
  hmgpu_status st = HMGPU_OK;
  std::vector<std::vector<int>> all(n);
  {
    ProfScope ps(c, K_H2D);              // (events on the compute stream: the staging itself runs beside it on the copy stream)
    for (int i = 0; i < n && st == HMGPU_OK; i++) {
      const hmgpu_picture_job& j = jobs[i];
      Picture& p = c->pics[j.pic];
      const hipStream_t hs = (i & 1) ? c->copy_stream2 : c->copy_stream;       // two copy lanes: two DMA engines
      wait_for_last_use(c, p, hs);
      if (p.sao_applied) { p.sao_applied = false; p.dev.sao_applied = 0; for (int k = 0; k < 3; k++) c->h_finals[j.pic].p[k] = p.dev.rec[k];
                           HIP_TRY(c, h2d_small(c, c->d_finals + j.pic, &c->h_finals[j.pic], sizeof(PlaneSet), hs)); }
      bool wp = false;
      { HostTimer ts(c, 1);
      for (int k = 0; k < j.num_slices && st == HMGPU_OK; k++) {
        st = register_slice(c, j.pic, k, j.slices[k], hs);
        all[i].push_back(k);
        wp |= j.slices[k] && j.slices[k]->weighted_pred != 0;
      }
      }
      SliceCall call;
      HostTimer ti(c, 2);
      if (st == HMGPU_OK) st = stage_inputs(c, j.pic, 0, all[i], wp, j.meta, j.coeffs, 0, c->num_ctus, hs, &call);
      if (st != HMGPU_OK) break;
      b.pic[i] = j.pic; b.first_ctu[i] = 0; b.num_ctus[i] = c->num_ctus;
      any_intra |= call.intra; any_wp |= call.wp; any_cells |= call.cells; any_bi |= call.bi; any_islice |= call.islice;
    }
    if (n > 1) {                        // (also after an error: whatever the second lane was given is ordered in front of the next event of the first)
      (void)hipEventRecord(c->copy_join, c->copy_stream2);
      (void)hipStreamWaitEvent(c->copy_stream, c->copy_join, 0);
    }
    if (st != HMGPU_OK) return st;
    c->copy_seq++;
    for (int i = 0; i < n; i++)
      if (const hmgpu_staging* sb = staging_of(c, jobs[i].meta, jobs[i].coeffs)) { const_cast<hmgpu_staging*>(sb)->copy_seq = c->copy_seq; const_cast<hmgpu_staging*>(sb)->reader = c; }
    HIP_TRY(c, hipEventRecord(c->copy_ev[c->copy_seq % 8], c->copy_stream));
    HIP_TRY(c, hipStreamWaitEvent(c->stream, c->copy_ev[c->copy_seq % 8], 0));
  }
  HostTimer tr(c, 3);
  for (int i = 0; i < n && st == HMGPU_OK; i++) st = extend_refs_of(c, jobs[i].pic, all[i]);
  if (st == HMGPU_OK) st = run_recon(c, b, any_intra, any_wp, any_cells, any_bi, any_islice);
  mark_use(c, b);
  return st;
}

hmgpu_status hmgpu_filter_pictures(hmgpu_ctx* c, int32_t n, const hmgpu_filter_job* jobs) {
  if (!c || !jobs || n < 1 || n > kMaxBatch) return HMGPU_EINVAL;
  for (int i = 0; i < n; i++) {
    if (!valid_pic(c, jobs[i].pic) || !jobs[i].pp) return HMGPU_EINVAL;
    if (jobs[i].pp->sao_enabled && !jobs[i].sao) return HMGPU_EINVAL;
    for (int k = 0; k < i; k++) if (jobs[k].pic == jobs[i].pic) return HMGPU_EINVAL;
  }
  hipSetDevice(c->device);
  Batch b; memset(&b, 0, sizeof(b));
  b.n = n;
  { HostTimer tsao(c, 4);
  for (int i = 0; i < n; i++) {
    Picture& p = c->pics[jobs[i].pic];
    p.sao_any = false;
    if (jobs[i].pp->sao_enabled) {
      std::vector<uint16_t> sidx = p.h_slice_idx, tidx = p.h_tile_idx;
      sidx.resize(c->num_ctus, 0);
      tidx.resize(c->num_ctus, 0);
      hmgpu_status st = stage_sao(c, p, jobs[i].pp, jobs[i].sao, sidx, tidx);
      if (st != HMGPU_OK) return st;
    }
    p.filter_ready = true;
    b.pic[i] = jobs[i].pic; b.first_ctu[i] = 0; b.num_ctus[i] = c->num_ctus;
  }
  }
  HostTimer tf(c, 5);
  hmgpu_status st = run_filter(c, b, 7);
  if (st != HMGPU_OK) return st;
  for (int i = 0; i < n && st == HMGPU_OK; i++) {
    Picture& p = c->pics[jobs[i].pic];
    if (p.sao_any) {
      p.sao_applied = true; p.dev.sao_applied = 1;
      st = push_final(c, jobs[i].pic);
      if (st == HMGPU_OK) st = push_picdev(c, jobs[i].pic);
    }
    p.extended = true;                   // (the batched border extension below)
  }
  if (st != HMGPU_OK) return st;
  { ProfScope ps(c, K_EXTEND); launch_extend(c->d_pics, b, c->seq.width, c->seq.height, c->mx[0], c->my[0], c->csx, c->csy, c->stream); }
  HIP_TRY(c, hipGetLastError());
  mark_use(c, b);
  return HMGPU_OK;
}

hmgpu_status hmgpu_filter_picture_stages(hmgpu_ctx* c, hmgpu_pic cur, const hmgpu_pic_params* pp, const hmgpu_sao_param* sao,
                                         int32_t stages) {
  if (!c || !valid_pic(c, cur) || !pp) return HMGPU_EINVAL;
  if ((stages & 4) && pp->sao_enabled && !sao) return HMGPU_EINVAL;
  hipSetDevice(c->device);
  Picture& p = c->pics[cur];
  p.sao_any = false;
  if ((stages & 4) && pp->sao_enabled) {
    // slice / tile index per CTU as handed over with the slices (host mirrors: no device round trip, the stream keeps running)
    std::vector<uint16_t> sidx = p.h_slice_idx, tidx = p.h_tile_idx;
    sidx.resize(c->num_ctus, 0);
    tidx.resize(c->num_ctus, 0);
    hmgpu_status st = stage_sao(c, p, pp, sao, sidx, tidx);
    if (st != HMGPU_OK) return st;
  }
  p.filter_ready = true;
  Batch b; memset(&b, 0, sizeof(b));
  b.n = 1; b.pic[0] = cur; b.first_ctu[0] = 0; b.num_ctus[0] = c->num_ctus;
  hmgpu_status st = run_filter(c, b, stages);
  if (st != HMGPU_OK) return st;
  if ((stages & 4) && p.sao_any) {
    // SAOProcess ran: the SAO planes are the picture now (HM: resYuv written in place after the snapshot copy)
    p.sao_applied = true; p.dev.sao_applied = 1;
    st = push_final(c, cur);
    if (st == HMGPU_OK) st = push_picdev(c, cur);
    if (st != HMGPU_OK) return st;
  }
  p.extended = false;
  st = ensure_extended(c, cur);          // the finished picture is ready to be referenced
  commit_use(c);
  return st;
}

hmgpu_status hmgpu_filter_picture(hmgpu_ctx* c, hmgpu_pic cur, const hmgpu_pic_params* pp, const hmgpu_sao_param* sao) {
  return hmgpu_filter_picture_stages(c, cur, pp, sao, 7);
}

hmgpu_status hmgpu_replay_batch(hmgpu_ctx* c, const hmgpu_pic* pics, int32_t n, int32_t stages, int32_t iters) {
  if (!c || !pics || n < 1 || n > kMaxBatch || iters < 0) return HMGPU_EINVAL;
  size_t ncalls = 0;
  for (int i = 0; i < n; i++) {
    if (!valid_pic(c, pics[i])) return HMGPU_EINVAL;
    if (i == 0) ncalls = c->pics[pics[i]].calls.size();
    else if (c->pics[pics[i]].calls.size() != ncalls) return HMGPU_EINVAL;
  }
  if ((stages & 8) && ncalls == 0) return HMGPU_EINVAL;
  hipSetDevice(c->device);
  // Two lanes: the batch is cut in two halves that run the same kernel sequence on two streams.  The halves are independent
  // pictures, so the lanes drift apart and kernels of different kinds (latency-bound motion compensation, ALU-heavier filters)
  // share the chip -- measured +12..25 % pictures/s over one lane.  One lane while profiling: per-kernel event
  // times are only meaningful when a kernel has the chip to itself.
  const int lanes = (c->replay_streams == 2 && n >= 2 && !c->profiling) ? 2 : 1;
  hipStream_t const main_stream = c->stream;
  if (lanes == 2) {                       // the second lane starts after everything enqueued so far on the context's stream
    HIP_TRY(c, hipEventRecord(c->lane_ev[0], main_stream));
    HIP_TRY(c, hipStreamWaitEvent(c->stream2, c->lane_ev[0], 0));
  }
  hmgpu_status result = HMGPU_OK;
  for (int it = 0; it < iters && result == HMGPU_OK; it++) {
    for (int lane = 0; lane < lanes && result == HMGPU_OK; lane++) {
      const int lo = lanes == 2 ? (lane == 0 ? 0 : n / 2) : 0, hi = lanes == 2 ? (lane == 0 ? n / 2 : n) : n;
      c->stream = lane == 0 ? main_stream : c->stream2;      // every launcher below enqueues on c->stream
      if (stages & 8) {
        for (size_t k = 0; k < ncalls && result == HMGPU_OK; k++) {
          Batch b; memset(&b, 0, sizeof(b));
          b.n = hi - lo;
          bool any_intra = false, any_wp = false, any_cells = false, any_bi = false, any_islice = false;
          for (int i = lo; i < hi; i++) {
            const SliceCall& sc = c->pics[pics[i]].calls[k];
            b.pic[i - lo] = pics[i]; b.first_ctu[i - lo] = sc.first_ctu; b.num_ctus[i - lo] = sc.num_ctus;
            any_intra |= sc.intra; any_wp |= sc.wp; any_cells |= sc.cells; any_bi |= sc.bi; any_islice |= sc.islice;
          }
          result = ensure_refs_extended(c, b, k);
          if (result == HMGPU_OK) result = run_recon(c, b, any_intra, any_wp, any_cells, any_bi, any_islice);
        }
      }
      if ((stages & 7) && result == HMGPU_OK) {
        Batch b; memset(&b, 0, sizeof(b));
        b.n = hi - lo;
        for (int i = lo; i < hi; i++) { b.pic[i - lo] = pics[i]; b.first_ctu[i - lo] = 0; b.num_ctus[i - lo] = c->num_ctus; }
        result = run_filter(c, b, stages & 7);
        if (result == HMGPU_OK) {
          { ProfScope ps(c, K_EXTEND); launch_extend(c->d_pics, b, c->seq.width, c->seq.height, c->mx[0], c->my[0], c->csx, c->csy, c->stream); }
          if (hipGetLastError() != hipSuccess) result = HMGPU_EDEVICE;
        }
      }
    }
  }
  c->stream = main_stream;
  if (lanes == 2) {                       // whatever follows on the context's stream (sync, download) sees both lanes finished
    HIP_TRY(c, hipEventRecord(c->lane_ev[1], c->stream2));
    HIP_TRY(c, hipStreamWaitEvent(main_stream, c->lane_ev[1], 0));
  }
  for (int i = 0; i < n; i++) touch(c, pics[i]);
  commit_use(c);                          // (the references were named by ensure_refs_extended)
  return result;
}

hmgpu_status hmgpu_set_streams(hmgpu_ctx* c, int32_t n) {
  if (!c || n < 1 || n > 2) return HMGPU_EINVAL;
  c->replay_streams = n;
  return HMGPU_OK;
}

hmgpu_status hmgpu_replay(hmgpu_ctx* c, hmgpu_pic cur, int32_t stages, int32_t iters) { return hmgpu_replay_batch(c, &cur, 1, stages, iters); }

hmgpu_status hmgpu_set_profiling(hmgpu_ctx* c, int32_t enable) {
  if (!c) return HMGPU_EINVAL;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  prof_drain(c);
  c->profiling = enable != 0;
  return HMGPU_OK;
}

hmgpu_status hmgpu_get_stats(hmgpu_ctx* c, hmgpu_stats* out, int32_t reset) {
  if (!c || !out) return HMGPU_EINVAL;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  prof_drain(c);
  memset(out, 0, sizeof(*out));
  for (int k = 0; k < HMGPU_NUM_KERNELS; k++) { out->kernel_ms[k] = c->kernel_ms[k]; out->kernel_launches[k] = c->kernel_launches[k]; }
  for (Picture& p : c->pics) {
    if (!p.in_use) continue;
    unsigned long long st[2 * kTuShards];
    HIP_TRY(c, hipMemcpy(st, p.dev.stats, sizeof(st), hipMemcpyDeviceToHost));
    for (int s = 0; s < kTuShards; s++) { out->intra_partitions += st[s]; out->inter_partitions += st[kTuShards + s]; }
    if (reset) HIP_TRY(c, hipMemset(p.dev.stats, 0, sizeof(st)));
  }
  if (reset) for (int k = 0; k < HMGPU_NUM_KERNELS; k++) { c->kernel_ms[k] = 0; c->kernel_launches[k] = 0; }
  return HMGPU_OK;
}

// ---------------------------------------------------------------------------------------------- finer seams
hmgpu_status hmgpu_inverse_transform_batch(hmgpu_ctx* c, int32_t log2_size, int32_t bit_depth, int32_t n, const int16_t* levels,
                                           const int8_t* qp_per, const int8_t* qp_rem, const uint8_t* flags, int16_t* resid) {
  if (!c || log2_size < 2 || log2_size > 5 || n < 1 || !levels || !qp_per || !qp_rem || !flags || !resid) return HMGPU_EINVAL;
  if (bit_depth < 8 || bit_depth > 12) return HMGPU_EUNSUPPORTED;
  hipSetDevice(c->device);
  const size_t elems = (size_t)n << (2 * log2_size);
  int16_t *d_lev = nullptr, *d_res = nullptr; int8_t *d_per = nullptr, *d_rem = nullptr; uint8_t* d_fl = nullptr;
  hmgpu_status st = HMGPU_OK;
  auto fail = [&](hipError_t e) { if (e != hipSuccess && st == HMGPU_OK) { c->last_err = (int32_t)e; st = HMGPU_EDEVICE; } };
  fail(hipMalloc((void**)&d_lev, elems * 2)); fail(hipMalloc((void**)&d_res, elems * 2));
  fail(hipMalloc((void**)&d_per, n)); fail(hipMalloc((void**)&d_rem, n)); fail(hipMalloc((void**)&d_fl, n));
  if (st == HMGPU_OK) {
    fail(hipMemcpy(d_lev, levels, elems * 2, hipMemcpyHostToDevice));
    fail(hipMemcpy(d_per, qp_per, n, hipMemcpyHostToDevice)); fail(hipMemcpy(d_rem, qp_rem, n, hipMemcpyHostToDevice));
    fail(hipMemcpy(d_fl, flags, n, hipMemcpyHostToDevice));
  }
  if (st == HMGPU_OK) {
    launch_itx_flat(log2_size, bit_depth, n, d_lev, d_per, d_rem, d_fl, d_res, c->stream);
    fail(hipGetLastError());
    fail(hipStreamSynchronize(c->stream));
    fail(hipMemcpy(resid, d_res, elems * 2, hipMemcpyDeviceToHost));
  }
  hipFree(d_lev); hipFree(d_res); hipFree(d_per); hipFree(d_rem); hipFree(d_fl);
  return st;
}

hmgpu_status hmgpu_mc_batch(hmgpu_ctx* c, int32_t is_chroma, int32_t bit_depth, const int16_t* ref_plane, int32_t ref_stride,
                            int32_t ref_w, int32_t ref_h, int32_t n, const int32_t* blocks, int32_t bi, int16_t* dst) {
  if (!c || !ref_plane || !blocks || !dst || n < 1 || ref_w < 1 || ref_h < 1 || ref_stride < ref_w) return HMGPU_EINVAL;
  if (bit_depth < 8 || bit_depth > 12) return HMGPU_EUNSUPPORTED;
  hipSetDevice(c->device);
  std::vector<int32_t> off(n);
  size_t total = 0;
  for (int i = 0; i < n; i++) {
    const int w = blocks[i * 6 + 2], h = blocks[i * 6 + 3];
    if (w < 2 || h < 2 || (w & 1) || (h & 1) || w > 64 || h > 64) return HMGPU_EINVAL;
    off[i] = (int32_t)total; total += (size_t)w * h;
  }
  // device copy of the plane with replicated margins (what extendPicBorder gives HM's xPredInterBlk); blocks and MVs must
  // keep the filter window within 96 samples of the plane
  const int M = 96;
  for (int i = 0; i < n; i++) {
    const int sh = is_chroma ? 3 : 2;
    const int bx = blocks[i * 6 + 0] + (blocks[i * 6 + 4] >> sh), by = blocks[i * 6 + 1] + (blocks[i * 6 + 5] >> sh);
    if (bx < -(M - 8) || by < -(M - 8) || bx + blocks[i * 6 + 2] > ref_w + M - 8 || by + blocks[i * 6 + 3] > ref_h + M - 8) return HMGPU_EINVAL;
  }
  const int pitch = (int)align_up((size_t)ref_w + 2 * M, 64) + 64;
  const int prow = ref_h + 2 * M;
  std::vector<int16_t> padded((size_t)pitch * prow, 0);
  for (int y = 0; y < prow; y++) {
    const int16_t* src = ref_plane + (size_t)std::min(std::max(y - M, 0), ref_h - 1) * ref_stride;
    int16_t* dstrow = padded.data() + (size_t)y * pitch;
    for (int x = 0; x < ref_w + 2 * M; x++) dstrow[x] = src[std::min(std::max(x - M, 0), ref_w - 1)];
  }
  int16_t *d_ref = nullptr, *d_dst = nullptr; int32_t *d_blk = nullptr, *d_off = nullptr;
  hmgpu_status st = HMGPU_OK;
  auto fail = [&](hipError_t e) { if (e != hipSuccess && st == HMGPU_OK) { c->last_err = (int32_t)e; st = HMGPU_EDEVICE; } };
  fail(hipMalloc((void**)&d_ref, padded.size() * 2)); fail(hipMalloc((void**)&d_dst, total * 2));
  fail(hipMalloc((void**)&d_blk, (size_t)n * 24)); fail(hipMalloc((void**)&d_off, (size_t)n * 4));
  if (st == HMGPU_OK) {
    fail(hipMemcpy(d_ref, padded.data(), padded.size() * 2, hipMemcpyHostToDevice));
    fail(hipMemcpy(d_blk, blocks, (size_t)n * 24, hipMemcpyHostToDevice));
    fail(hipMemcpy(d_off, off.data(), (size_t)n * 4, hipMemcpyHostToDevice));
  }
  if (st == HMGPU_OK) {
    launch_mc_flat(is_chroma, bit_depth, d_ref + (size_t)M * pitch + M, pitch, ref_w, ref_h, n, d_blk, d_off, bi, d_dst, c->stream);
    fail(hipGetLastError());
    fail(hipStreamSynchronize(c->stream));
    fail(hipMemcpy(dst, d_dst, total * 2, hipMemcpyDeviceToHost));
  }
  hipFree(d_ref); hipFree(d_dst); hipFree(d_blk); hipFree(d_off);
  return st;
}

}  // extern "C"
