// vf_api.hip -- the C ABI of libveritasfi_hip.so (include/veritasfi_hip.h): handle management,
// path selection and stream orchestration around the kernels in vf_kernels.hip.
//
// Reference surface replaced: faiss.IndexFlatIP build + search behind FaissRetriever
// (src/utils/faissRetriever.py:11-38), sklearn.cosine_similarity + argsort
// (experiments/retriever/step3_mul.py:233-289), compute_similarity_mtx
// (src/utils/ensembleRetriever.py:275-279), rank_chunk fusion (src/utils/vllmManager.py:454-457).
#include "../../include/veritasfi_hip.h"
#include "vf_internal.h"

#include <fcntl.h>
#include <float.h>
#include <sys/stat.h>
#include <unistd.h>
#include <math.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <string>
#include <vector>

using namespace vf;

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_err;

static int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
int vf::set_error(int code, const std::string& msg) { return fail(code, msg); }

#define VF_HIP(expr)                                                                                   \
    do {                                                                                               \
        hipError_t _e = (expr);                                                                        \
        if (_e != hipSuccess)                                                                          \
            return fail(VF_EHIP, std::string(#expr) + ": " + hipGetErrorString(_e) + " (" + __FILE__ + \
                                     ":" + std::to_string(__LINE__) + ")");                            \
    } while (0)

#define VF_TRY(expr)             \
    do {                         \
        int _rc = (expr);        \
        if (_rc != VF_OK) return _rc; \
    } while (0)

extern "C" int vf_version(void) { return VF_VERSION; }
extern "C" const char* vf_last_error(void) { return g_err.c_str(); }

extern "C" int vf_device_count(int32_t* out) {
    if (!out) return fail(VF_EINVAL, "vf_device_count: null out");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *out = 0; return fail(VF_EHIP, std::string("hipGetDeviceCount: ") + hipGetErrorString(e)); }
    *out = n;
    return VF_OK;
}

// ------------------------------------------------------------------------------------------------
// handle
// ------------------------------------------------------------------------------------------------
namespace {

constexpr int kSlots = 4;

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    int ensure(size_t need) {
        if (need <= bytes) return VF_OK;
        if (p) { (void)hipFree(p); p = nullptr; bytes = 0; }
        hipError_t e = hipMalloc(&p, need);
        if (e != hipSuccess) return fail(VF_ENOMEM, std::string("hipMalloc(") + std::to_string(need) + "): " + hipGetErrorString(e));
        bytes = need;
        return VF_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
    template <class T> T* as() const { return (T*)p; }
};

struct HostBuf {   // pinned host staging
    void* p = nullptr;
    size_t bytes = 0;
    int ensure(size_t need) {
        if (need <= bytes) return VF_OK;
        if (p) { (void)hipHostFree(p); p = nullptr; bytes = 0; }
        hipError_t e = hipHostMalloc(&p, need, hipHostMallocDefault);
        if (e != hipSuccess) return fail(VF_ENOMEM, std::string("hipHostMalloc(") + std::to_string(need) + "): " + hipGetErrorString(e));
        bytes = need;
        return VF_OK;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; bytes = 0; }
};

struct Slot {
    hipStream_t stream = nullptr;        // everything but the main scan (any CU)
    hipStream_t scan_stream = nullptr;   // the main k_scan launches: CU-masked to the scan partition when the split is on, else == stream
    hipEvent_t ev_pro = nullptr;         // this slot's prologue (queries, sample pass, threshold seed) is done
    hipEvent_t ev_in = nullptr, ev_done = nullptr, ev_scan = nullptr;  // ev_scan: after this slot's main k_scan
    hipEvent_t ev_t[4] = {nullptr, nullptr, nullptr, nullptr};  // profile: scan begin/end, pipeline begin/end
    bool timed = false;
    int wide_launches = 0, wide_queries = 0;   // k_scan_wide main passes of the pending search / queries they served
    int scan_kernel = 0;                       // main-scan kernel of the pending search: 1 k_scan, 2 k_scan2, 3 k_scan_wide
    DevBuf qn, qimg, s0, cnt, tau, hist, hist_coarse, cand, flags, counts;   // fused-path state
    DevBuf dbg, wgbase, tilecnt, sib;
    DevBuf qimg8, epsq;                  // k_scan_wide8: hi / lo e4m3 query image, per-query certificate bounds
    int64_t wgbase_n = -1; int wgbase_grid = -1;
    DevBuf dense_s, cn_tmp, parts_ids, parts_sc, run_ids, run_sc, qsel; // exact-path scratch
    int* h_flags = nullptr;      // pinned, [max batches * 64]
    u32* h_counts = nullptr;     // pinned
    int* d_flags = nullptr;      // device views of the two above
    u32* d_counts = nullptr;
    size_t h_cap = 0;
    // pending call
    bool pending = false;
    const float* d_queries = nullptr;
    int nq = 0, k = 0, path = 0;
    int64_t* d_ids = nullptr;
    float* d_scores = nullptr;
    hipStream_t user_stream = nullptr;
};

// per-slot state of a group search: one feed stream / staging set per shard, on the shard's device
struct GroupSlot {
    std::vector<hipStream_t> feed;       // [G] on shard g's device: query copy -> shard search -> result copy home
    std::vector<hipEvent_t> ev;          // [G] result of shard g has landed in `all` on the home device
    std::vector<DevBuf> dq, blob;        // [G] queries / packed result [ids | scores] on shard g's device
    DevBuf all;                          // home device: G packed parts, the merge kernel's input
    hipEvent_t ev_in = nullptr;          // home device: the caller's stream at _begin time
    // Host-staged exchange for shards WITHOUT peer access to the home device (hipDeviceEnablePeerAccess refused, or the test hook
    // vf_debug_force_no_peer): queries home -> pinned host -> shard, packed result shard -> pinned host -> home.  Every copy is issued
    // on a stream of the device that owns the DEVICE side of it; events carry the order across devices.  Created on first use.
    hipStream_t hq_stream = nullptr;     // home device: the query copy to the host (once per search, shared by the staged shards)
    hipEvent_t ev_hq = nullptr;          // home device: the queries are in `hq`
    HostBuf hq;                          // pinned: the batch's queries
    std::vector<hipStream_t> hfeed;      // [G] home device: shard g's result, host -> `all`
    std::vector<hipEvent_t> ev_hb;       // [G] shard device: shard g's result is in hb[g]
    std::vector<hipEvent_t> ev_home;     // [G] home device: ... and has landed in `all` (what the merge waits for instead of ev[g])
    std::vector<HostBuf> hb;             // [G] pinned: shard g's packed result
    std::vector<char> staged_last;       // [G] the last search of this slot took the staged way for shard g (ev_home[g] is recorded)
    bool pending = false;
    int nq = 0, k = 0;
    int64_t* d_ids = nullptr; float* d_scores = nullptr;
    hipStream_t user = nullptr;
};

}  // namespace

struct vf_index {
    int device = 0;
    int64_t n = 0;
    int d = 0, dp = 0, dtype = 0;  // dtype: how rows are HELD in HBM (VF_DTYPE_F32 / _F16 / _FP8_E4M3)
    int64_t id_offset = 0;
    int n_cu = 256;
    int64_t aux_applied = -1;   // the CU split the first slot's streams were created with (-1: no slot yet); fixed from then on
    bool owns_rows = false;
    void* rows_orig = nullptr;        // as given (fp32 or fp16), [n][d]
    void* rows_scan = nullptr;    // fp16 [n][dp]; may alias rows_orig
    bool owns_scan = false;
    float* norm = nullptr;            // canonical norms [n]
    float* inv_scan = nullptr;        // [n]
    float* cn_cache = nullptr;        // canonical normalised rows when n <= kSmallN
    std::mutex mu;
    Slot slots[kSlots];
    // options
    int64_t force_path = -1, sample_rows = -1, margin = -1, cap_opt = 0, waves_opt = 0, scan_g = 0,
            refresh_every = 128, debug = 0, steal_opt = 0, wide_opt = 1, wide_sync = -1, wide_mfma = -1, wide8_waves = 8, wide8_stage = 0,
            aux_cus = -1, sample_grid = -1, overlap_scans = -1, scan_impl = 2, sample_impl = -1;   // aux_cus / overlap_scans: -1 = auto (resolved_split)   // scan_impl: 1 = k_scan (register loads), 2 = k_scan2 (whole-line LDS-DMA) where it fits   // aux_cus: CUs the main scan leaves to the small kernels of the other slots (0 = no split)   // wide_sync: -1 siblings of a wide row group run free (default: fastest), >= 0 = the slack in super-tiles  // steal_opt: cross-workgroup tile pool in the main scan (measured slower: DESIGN.md 5)  // wide_opt: 0 never, 1 auto (nq >= 129), > 1 = from that many queries
    vf_search_stats stats{};
    bool profile = false;
    double prof_scan_ms = 0.0, prof_pipe_ms = 0.0;
    int64_t prof_launches = 0, prof_bytes = 0;
    hipEvent_t ev_span = nullptr;     // profile: the first timed main scan since the option was set began here (scan stream)
    bool span_started = false;
    bool span_slot[4] = {false, false, false, false};   // slots whose ev_t[1] (end of their latest timed scan) belongs to this span
    // host-buffer entry (vf_index_search): per-handle device staging, grown on demand, reused across calls
    DevBuf st_q, st_ids, st_sc;
    // ---- group handle (vf_index_create_sharded / vf_index_group): the corpus is row-sharded over `shards`, one per
    // device; this handle owns them.  device = the HOME device: queries arrive there and the merged result lands there.
    std::vector<vf_index*> shards;
    std::vector<int> peer_ok;         // [G] 1 = home <-> shard g peer access is on in both directions (or same device)
    GroupSlot gslots[kSlots];
};

// flags / candidate counts are written by k_final straight into host-mapped pinned memory
static int ensure_pinned(Slot& s, size_t nq_total) {
    if (nq_total <= s.h_cap) return VF_OK;
    if (s.h_flags) (void)hipHostFree(s.h_flags);
    if (s.h_counts) (void)hipHostFree(s.h_counts);
    s.h_flags = nullptr; s.h_counts = nullptr; s.h_cap = 0;
    size_t cap = std::max<size_t>(nq_total, 256);
    VF_HIP(hipHostMalloc((void**)&s.h_flags, cap * sizeof(int), hipHostMallocMapped));
    VF_HIP(hipHostMalloc((void**)&s.h_counts, cap * sizeof(u32), hipHostMallocMapped));
    VF_HIP(hipHostGetDevicePointer((void**)&s.d_flags, s.h_flags, 0));
    VF_HIP(hipHostGetDevicePointer((void**)&s.d_counts, s.h_counts, 0));
    s.h_cap = cap;
    return VF_OK;
}

static int build_common(vf_index* ix) {
    hipDeviceProp_t prop;
    VF_HIP(hipGetDeviceProperties(&prop, ix->device));
    ix->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    VF_HIP(scan_configure());
    const int dt = ix->dtype;
    const size_t scan_esz = dt == VF_DTYPE_FP8_E4M3 ? 1 : 2;  // fp8 rows are scanned as bytes, everything else as fp16
    ix->dp = (ix->d + 127) / 128 * 128;  // whole 128-element segment pairs: see pick_G
    const size_t npad = (size_t)ix->n + 64;
    VF_HIP(hipMalloc((void**)&ix->norm, npad * sizeof(float)));
    VF_HIP(hipMalloc((void**)&ix->inv_scan, npad * sizeof(float)));
    VF_HIP(hipMemset(ix->norm, 0, npad * sizeof(float)));
    VF_HIP(hipMemset(ix->inv_scan, 0, npad * sizeof(float)));
    const bool need_scan = ix->n > kSmallN;  // small corpora never run the fused scan
    void* scan_out = nullptr;
    if (need_scan) {
        if (dt != VF_DTYPE_F32 && ix->dp == ix->d && ((uintptr_t)ix->rows_orig % 16) == 0) {
            ix->rows_scan = ix->rows_orig;  // fp16 / fp8 rows of a whole number of segments are scanned in place
            ix->owns_scan = false;
        } else {
            VF_HIP(hipMalloc((void**)&ix->rows_scan, (size_t)ix->n * ix->dp * scan_esz));
            ix->owns_scan = true;
            scan_out = ix->rows_scan;
        }
    }
    VF_HIP(launch_prep_rows(ix->rows_orig, dt, ix->n, ix->d, ix->dp, scan_out, ix->norm, ix->inv_scan, nullptr));
    if (ix->n > 0 && ix->n <= kSmallN) {
        VF_HIP(hipMalloc((void**)&ix->cn_cache, (size_t)ix->n * ix->d * sizeof(float)));
        VF_HIP(launch_normalize_rows(ix->rows_orig, dt, 0, ix->n, ix->d, ix->norm, ix->cn_cache, nullptr));
    }
    VF_HIP(hipDeviceSynchronize());
    return VF_OK;
}

// A slot's stream and events are created on its first use: a one-shot index (the reference builds one per
// select_top_chunks call, step3_mul.py:233-253) only ever touches slot 0.
// CU split and scan overlap, resolved.  Auto (-1): shards of up to 6M rows run their main scans on all but 32 CUs (one
// per shader engine: a mask that takes CUs from only some SEs leaves those SEs with more workgroups than CUs -- the
// dispatcher hands every SE the same number -- and a scan then takes two rounds; tools/ubench/cu_mask_probe.hip) and let
// consecutive scans overlap; larger shards keep the whole chip and ordered scans (measured, round 3: 1.25M rows 0.384 ->
// 0.362 ms per batch, 2.5M 0.717 -> 0.682, 5M 1.280 -> 1.269, 10M no change; profiles/r03_scan2_sweep.log).
constexpr int64_t kSplitMaxRows = 6'000'000;
constexpr int64_t kScan2rMinRows = 1'100'000;   // k_scan2r (where its shapes exist) above this many rows: below, the workgroup's longer start costs more than the ring gains
// Round 6: where k_scan2r serves the rows (fp16 rows of 384 / 512 / 768 / 1024 elements; a wave keeps 16-24 KB in flight there) the split + overlapping scans
// win at EVERY size -- 7.5M rows 1.878 -> 1.811 ms per batch, 10M rows 2.538 -> 2.466-2.473 (0.7585 -> 0.78 of 8 TB/s), whole chip +
// ordered scans with k_scan2 being the 2.538; k_scan2r on the whole chip with ordered scans LOSES (2.58-2.59): profiles/r06_scan2r_10m.log
// the widths on which k_scan2r was MEASURED against the kernel it replaces and is the default (its other shapes: scan_impl = 5)
// fp16 rows of 1024 / 512 / 384 elements (round 6, profiles/r06_scan2r_fp16_other_widths_ab.log; k_scan2r + its sample pass on the CU split
// with overlapping scans against the default before): 8M x 1024 2.725 -> 2.63-2.65 ms per batch (0.754 -> 0.774-0.781 of 8 TB/s; k_scan served
// that width: k_scan2's image does not fit), 1.25M x 1024 0.465-0.467 -> 0.430-0.444, 10M x 512 1.711 -> 1.620-1.643 (0.765 -> 0.785-0.796),
// 10M x 384 1.330 -> 1.262, 1.25M x 512 0.266-0.270 -> 0.260-0.265
static bool scan2r_auto_width(int dp, bool f8) { return f8 ? (dp == 768 || dp == 1024) : (dp == 768 || dp == 1024 || dp == 512 || dp == 384); }
static int64_t split_limit(const vf_index* ix) {
    const bool r_rows = ix->dtype != VF_DTYPE_FP8_E4M3 && ix->scan_impl != 4 && ix->scan_impl != 1 && ix->scan_impl != 3 && !ix->steal_opt &&
                        scan2r_auto_width(ix->dp, false) && scan2r_stage_cap(ix->dp, kMaxBatch, 0) >= 256;
    return r_rows ? INT64_MAX : kSplitMaxRows;
}
static int64_t resolved_aux(const vf_index* ix) {
    if (ix->aux_applied >= 0) return ix->aux_applied;   // what the existing scan streams are masked with (0 if masking failed)
    int64_t a = ix->aux_cus >= 0 ? ix->aux_cus : (ix->n <= split_limit(ix) ? 32 : 0);
    if (a <= 0 || ix->n_cu < 64 || a * 2 > ix->n_cu) return 0;
    return a;
}
static bool resolved_overlap(const vf_index* ix) {
    return ix->overlap_scans >= 0 ? ix->overlap_scans != 0 : resolved_aux(ix) > 0;
}

static int ensure_slot(vf_index* ix, Slot& s) {
    if (s.stream) return VF_OK;
    VF_HIP(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
    // CU partition ("small shards"): a main-scan workgroup owns its CU (234 VGPRs x 8 waves, 128 KB LDS), so the
    // prologue and k_final of the OTHER slots used to queue behind it.  The main scans run on a stream whose CU mask
    // leaves `aux_cus` CUs alone (mask bits interleave over the 8 XCDs: bits [0, n_cu - aux) = all but the last aux / 8 CUs
    // of every XCD; tools/ubench/cu_mask_probe.hip); everything else runs on an unmasked stream and finds those CUs free.
    s.scan_stream = s.stream;
    int64_t aux = resolved_aux(ix);
    if (aux) {
        const int words = (ix->n_cu + 31) / 32;
        std::vector<uint32_t> mask(words, 0u);
        for (int b = 0; b < ix->n_cu - (int)aux; ++b) mask[b / 32] |= 1u << (b % 32);
        hipStream_t ms = nullptr;
        if (hipExtStreamCreateWithCUMask(&ms, (uint32_t)words, mask.data()) == hipSuccess && ms) s.scan_stream = ms;
        else { (void)hipGetLastError(); if (ix->aux_applied < 0) aux = 0; }   // no CU masking on this stack: one stream, whole-chip grids
    }
    // the split is a property of the streams: plans, stats and the overlap decision use what was APPLIED here, whatever the
    // option is set to later (vf_index_set_option rejects a change once slots exist)
    if (ix->aux_applied < 0) ix->aux_applied = aux;
    VF_HIP(hipEventCreateWithFlags(&s.ev_in, hipEventDisableTiming));
    VF_HIP(hipEventCreateWithFlags(&s.ev_done, hipEventDisableTiming));
    VF_HIP(hipEventCreateWithFlags(&s.ev_scan, hipEventDisableTiming));
    VF_HIP(hipEventCreateWithFlags(&s.ev_pro, hipEventDisableTiming));
    for (int e = 0; e < 4; ++e) VF_HIP(hipEventCreate(&s.ev_t[e]));
    return VF_OK;
}

static void destroy_index(vf_index* ix) {
    if (!ix) return;
    if (!ix->shards.empty()) {  // group handle: per-slot staging on every shard's device, then the shards themselves
        for (int i = 0; i < kSlots; ++i) {
            GroupSlot& gs = ix->gslots[i];
            for (size_t g = 0; g < ix->shards.size(); ++g) {
                (void)hipSetDevice(ix->shards[g]->device);
                (void)hipDeviceSynchronize();
                if (g < gs.dq.size()) gs.dq[g].release();
                if (g < gs.blob.size()) gs.blob[g].release();
                if (g < gs.feed.size() && gs.feed[g]) (void)hipStreamDestroy(gs.feed[g]);
                if (g < gs.ev.size() && gs.ev[g]) (void)hipEventDestroy(gs.ev[g]);
                if (g < gs.ev_hb.size() && gs.ev_hb[g]) (void)hipEventDestroy(gs.ev_hb[g]);
                if (g < gs.hb.size()) gs.hb[g].release();
            }
            (void)hipSetDevice(ix->device);
            gs.all.release();
            gs.hq.release();
            for (hipStream_t st : gs.hfeed) if (st) (void)hipStreamDestroy(st);
            for (hipEvent_t ev : gs.ev_home) if (ev) (void)hipEventDestroy(ev);
            if (gs.hq_stream) (void)hipStreamDestroy(gs.hq_stream);
            if (gs.ev_hq) (void)hipEventDestroy(gs.ev_hq);
            if (gs.ev_in) (void)hipEventDestroy(gs.ev_in);
        }
        for (vf_index* sh : ix->shards) destroy_index(sh);
        ix->shards.clear();
    }
    (void)hipSetDevice(ix->device);
    (void)hipDeviceSynchronize();
    ix->st_q.release(); ix->st_ids.release(); ix->st_sc.release();
    for (int i = 0; i < kSlots; ++i) {
        Slot& s = ix->slots[i];
        DevBuf* bufs[] = {&s.qn, &s.qimg, &s.s0, &s.cnt, &s.tau, &s.hist, &s.hist_coarse, &s.cand, &s.flags, &s.counts, &s.dense_s,
                          &s.cn_tmp, &s.parts_ids, &s.parts_sc, &s.run_ids, &s.run_sc, &s.qsel, &s.dbg, &s.wgbase, &s.tilecnt, &s.sib};
        for (DevBuf* b : bufs) b->release();
        if (s.h_flags) (void)hipHostFree(s.h_flags);
        if (s.h_counts) (void)hipHostFree(s.h_counts);
        if (s.scan_stream && s.scan_stream != s.stream) (void)hipStreamDestroy(s.scan_stream);
        if (s.stream) (void)hipStreamDestroy(s.stream);
        if (s.ev_pro) (void)hipEventDestroy(s.ev_pro);
        if (s.ev_in) (void)hipEventDestroy(s.ev_in);
        if (s.ev_done) (void)hipEventDestroy(s.ev_done);
        if (s.ev_scan) (void)hipEventDestroy(s.ev_scan);
        for (int e = 0; e < 4; ++e) if (s.ev_t[e]) (void)hipEventDestroy(s.ev_t[e]);
    }
    if (ix->owns_scan && ix->rows_scan) (void)hipFree(ix->rows_scan);
    if (ix->owns_rows && ix->rows_orig) (void)hipFree(ix->rows_orig);
    if (ix->norm) (void)hipFree(ix->norm);
    if (ix->inv_scan) (void)hipFree(ix->inv_scan);
    if (ix->cn_cache) (void)hipFree(ix->cn_cache);
    if (ix->ev_span) (void)hipEventDestroy(ix->ev_span);
    delete ix;
}

static int create_impl(vf_index** out, const void* rows, bool rows_on_device, int64_t n, int32_t d, int32_t dtype,
                       int32_t device_id, int64_t id_offset) {
    if (!out) return fail(VF_EINVAL, "vf_index_create: null out");
    *out = nullptr;
    if (n < 0 || d <= 0 || (n > 0 && !rows)) return fail(VF_EINVAL, "vf_index_create: bad rows/n/d");
    if (dtype != VF_DTYPE_F32 && dtype != VF_DTYPE_F16 && dtype != VF_DTYPE_FP8_E4M3)
        return fail(VF_EINVAL, "vf_index_create: unknown dtype");
    if (n >= (int64_t)0xFFFFFFFFll) return fail(VF_EUNSUPPORTED, "vf_index_create: more than 2^32-1 rows per shard");
    int ndev = 0;
    VF_HIP(hipGetDeviceCount(&ndev));
    if (device_id < 0 || device_id >= ndev) return fail(VF_EINVAL, "vf_index_create: bad device_id");
    VF_HIP(hipSetDevice(device_id));
    vf_index* ix = new (std::nothrow) vf_index();
    if (!ix) return fail(VF_ENOMEM, "vf_index_create: host allocation failed");
    ix->device = device_id; ix->n = n; ix->d = d; ix->id_offset = id_offset;
    ix->dtype = dtype;  // fp8 (OCP e4m3) rows stay fp8 in HBM: scanned as bytes, converted in registers
    const size_t esz = dtype == VF_DTYPE_F32 ? 4 : (dtype == VF_DTYPE_F16 ? 2 : 1);
    int rc = VF_OK;
    if (rows_on_device) {
        ix->rows_orig = const_cast<void*>(rows);
        ix->owns_rows = false;
    } else if (n > 0) {
        hipError_t e = hipMalloc(&ix->rows_orig, (size_t)n * d * esz);
        if (e != hipSuccess) rc = fail(VF_ENOMEM, std::string("hipMalloc(corpus): ") + hipGetErrorString(e));
        else {
            ix->owns_rows = true;
            e = hipMemcpy(ix->rows_orig, rows, (size_t)n * d * esz, hipMemcpyHostToDevice);
            if (e != hipSuccess) rc = fail(VF_EHIP, std::string("hipMemcpy(corpus): ") + hipGetErrorString(e));
        }
    }
    if (rc == VF_OK) rc = build_common(ix);
    if (rc != VF_OK) { std::string keep = g_err; destroy_index(ix); g_err = keep; return rc; }
    *out = ix;
    return VF_OK;
}

extern "C" int vf_index_create(vf_index** out, const void* rows, int64_t n, int32_t d, int32_t dtype,
                               int32_t device_id, int64_t id_offset) {
    DeviceGuard restore_callers_device;
    return create_impl(out, rows, false, n, d, dtype, device_id, id_offset);
}

extern "C" int vf_index_create_device(vf_index** out, const void* d_rows, int64_t n, int32_t d, int32_t dtype,
                                      int32_t device_id, int64_t id_offset) {
    DeviceGuard restore_callers_device;
    return create_impl(out, d_rows, true, n, d, dtype, device_id, id_offset);
}

// ------------------------------------------------------------------------------------------------
// Corpus file (.vfc): a 64-byte header followed by n * d elements, row-major.  Replaces "pull every
// embedding out of Chroma into Python lists at start-up" (src/utils/ensembleRetriever.py:39-43,
// src/utils/faissRetriever.py:14) for corpora that size cannot go through.
//   +0  char[8] "VFCORPUS"   +8 u32 version (1)   +12 u32 dtype (VF_DTYPE_*)   +16 u64 n   +24 u32 d
//   +28 u32 flags (bit 0: an int64[n] external-id table follows the rows)   +32 .. +63 reserved (0)
// ------------------------------------------------------------------------------------------------
struct VfcHeader {
    char magic[8];
    uint32_t version, dtype;
    uint64_t n;
    uint32_t d, flags;
    uint8_t reserved[32];
};
static_assert(sizeof(VfcHeader) == 64, "corpus file header is 64 bytes");

static int read_vfc_header(int fd, const char* path, VfcHeader* h, size_t* esz) {
    if (pread(fd, h, sizeof(*h), 0) != (ssize_t)sizeof(*h)) return fail(VF_EINVAL, std::string("corpus file too short: ") + path);
    if (memcmp(h->magic, "VFCORPUS", 8) != 0) return fail(VF_EINVAL, std::string("not a corpus file (bad magic): ") + path);
    if (h->version != 1) return fail(VF_EUNSUPPORTED, std::string("corpus file version not supported: ") + path);
    if (h->dtype > VF_DTYPE_FP8_E4M3 || h->d == 0) return fail(VF_EINVAL, std::string("corpus file header is corrupt: ") + path);
    *esz = h->dtype == VF_DTYPE_F32 ? 4 : (h->dtype == VF_DTYPE_F16 ? 2 : 1);
    struct stat st;
    if (fstat(fd, &st) != 0) return fail(VF_EINVAL, std::string("cannot stat corpus file: ") + path);
    // n * d * element size (+ the id table) against the file's size, WITHOUT wrapping: a header with n = 2^61 and d = 16 would
    // otherwise multiply to a small number, pass this check and hand a huge row count to the loader (round 5: found by the header
    // fuzz of tools/host_sanitize_check.cc; unsigned wrap-around is not an error any sanitizer reports)
    unsigned long long payload = 0, ids = 0, need = 0;
    if (__builtin_mul_overflow((unsigned long long)h->n, (unsigned long long)h->d * (unsigned long long)*esz, &payload) ||
        __builtin_mul_overflow((unsigned long long)h->n, (h->flags & 1u) ? 8ull : 0ull, &ids) ||
        __builtin_add_overflow(payload, ids, &need) || __builtin_add_overflow(need, 64ull, &need))
        return fail(VF_EINVAL, std::string("corpus file header is corrupt (n x d overflows): ") + path);
    if ((unsigned long long)st.st_size < need) return fail(VF_EINVAL, std::string("corpus file is truncated: ") + path);
    if (h->n >= 0x7fffffffffffffffull / 2) return fail(VF_EINVAL, std::string("corpus file header is corrupt: ") + path);
    return VF_OK;
}

extern "C" int vf_corpus_file_info(const char* path, int64_t* n, int32_t* d, int32_t* dtype, int32_t* has_ids) {
    if (!path) return fail(VF_EINVAL, "vf_corpus_file_info: null path");
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return fail(VF_EINVAL, std::string("cannot open corpus file: ") + path);
    VfcHeader h; size_t esz = 0;
    const int rc = read_vfc_header(fd, path, &h, &esz);
    close(fd);
    if (rc != VF_OK) return rc;
    if (n) *n = (int64_t)h.n;
    if (d) *d = (int32_t)h.d;
    if (dtype) *dtype = (int32_t)h.dtype;
    if (has_ids) *has_ids = (int32_t)(h.flags & 1u);
    return VF_OK;
}

// Rows [row_lo, row_hi) of the file become an index on `device_id` (a rank's shard: SURVEY.md 8e).  The rows are
// streamed through two pinned staging buffers (read() of one overlaps the H2D copy of the other).
extern "C" int vf_index_create_from_file(vf_index** out, const char* path, int64_t row_lo, int64_t row_hi,
                                         int32_t device_id, int64_t id_offset) {
    DeviceGuard restore_callers_device;
    if (!out || !path) return fail(VF_EINVAL, "vf_index_create_from_file: null argument");
    *out = nullptr;
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return fail(VF_EINVAL, std::string("cannot open corpus file: ") + path);
    VfcHeader h; size_t esz = 0;
    int rc = read_vfc_header(fd, path, &h, &esz);
    if (rc == VF_OK && (row_lo < 0 || row_hi < row_lo || (uint64_t)row_hi > h.n))
        rc = fail(VF_EINVAL, "vf_index_create_from_file: row range outside the file");
    int ndev = 0;
    if (rc == VF_OK && (hipGetDeviceCount(&ndev) != hipSuccess || device_id < 0 || device_id >= ndev))
        rc = fail(VF_EINVAL, "vf_index_create_from_file: bad device_id");
    if (rc != VF_OK) { close(fd); return rc; }
    const int64_t n = row_hi - row_lo;
    const size_t row_bytes = (size_t)h.d * esz, total = (size_t)n * row_bytes;
    void* d_rows = nullptr;
    char* stage[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    const size_t chunk = 64u << 20;
    hipError_t e = hipSetDevice(device_id);
    if (e == hipSuccess && total) e = hipMalloc(&d_rows, total);
    for (int i = 0; i < 2 && e == hipSuccess && total; ++i) {
        e = hipHostMalloc((void**)&stage[i], chunk, hipHostMallocDefault);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&ev[i], hipEventDisableTiming);
    }
    if (e != hipSuccess) rc = fail(VF_ENOMEM, std::string("vf_index_create_from_file: ") + hipGetErrorString(e));
    size_t done = 0;
    for (int i = 0; rc == VF_OK && done < total; i ^= 1) {
        const size_t len = std::min(chunk, total - done);
        if (hipEventSynchronize(ev[i]) != hipSuccess) { rc = fail(VF_EHIP, "vf_index_create_from_file: event wait failed"); break; }
        size_t got = 0;
        while (got < len) {
            const ssize_t r = pread(fd, stage[i] + got, len - got, (off_t)(64 + (size_t)row_lo * row_bytes + done + got));
            if (r <= 0) { rc = fail(VF_EINVAL, std::string("read error in corpus file: ") + path); break; }
            got += (size_t)r;
        }
        if (rc != VF_OK) break;
        e = hipMemcpyAsync((char*)d_rows + done, stage[i], len, hipMemcpyHostToDevice, nullptr);
        if (e == hipSuccess) e = hipEventRecord(ev[i], nullptr);
        if (e != hipSuccess) { rc = fail(VF_EHIP, std::string("vf_index_create_from_file: ") + hipGetErrorString(e)); break; }
        done += len;
    }
    close(fd);
    if (rc == VF_OK && hipDeviceSynchronize() != hipSuccess) rc = fail(VF_EHIP, "vf_index_create_from_file: copy failed");
    for (int i = 0; i < 2; ++i) {
        if (stage[i]) (void)hipHostFree(stage[i]);
        if (ev[i]) (void)hipEventDestroy(ev[i]);
    }
    if (rc == VF_OK) {
        rc = create_impl(out, d_rows, true, n, (int32_t)h.d, (int32_t)h.dtype, device_id, id_offset);
        // create_impl borrowed the device rows: hand them over to the index
        if (rc == VF_OK) { (*out)->owns_rows = true; d_rows = nullptr; }
    }
    if (d_rows) (void)hipFree(d_rows);
    return rc;
}

extern "C" int vf_index_destroy(vf_index* ix) {
    DeviceGuard restore_callers_device;
    if (!ix) return VF_OK;
    destroy_index(ix);
    return VF_OK;
}

extern "C" int vf_index_info(vf_index* ix, int64_t* n, int32_t* d, int32_t* dtype, int32_t* device_id) {
    if (!ix) return fail(VF_EINVAL, "vf_index_info: null handle");
    if (n) *n = ix->n;
    if (d) *d = ix->d;
    if (dtype) *dtype = ix->dtype;
    if (device_id) *device_id = ix->device;
    return VF_OK;
}

extern "C" int vf_index_stats(vf_index* ix, vf_search_stats* out) {
    if (!ix || !out) return fail(VF_EINVAL, "vf_index_stats: null argument");
    std::lock_guard<std::mutex> g(ix->mu);
    *out = ix->stats;
    return VF_OK;
}

extern "C" int vf_index_profile(vf_index* ix, double* scan_ms_total, int64_t* scan_launches, double* pipeline_ms_total,
                                int64_t* scan_bytes_per_launch) {
    if (!ix) return fail(VF_EINVAL, "vf_index_profile: null handle");
    std::lock_guard<std::mutex> g(ix->mu);
    double sm = ix->prof_scan_ms, pm = ix->prof_pipe_ms;
    int64_t nl = ix->prof_launches, nb = ix->prof_bytes;
    for (vf_index* sh : ix->shards) {  // group: totals over the shards (equal blocks: bytes per launch of the first)
        std::lock_guard<std::mutex> lk(sh->mu);
        sm += sh->prof_scan_ms; pm += sh->prof_pipe_ms; nl += sh->prof_launches;
        if (nb == 0) nb = sh->prof_bytes;
    }
    if (scan_ms_total) *scan_ms_total = sm;
    if (scan_launches) *scan_launches = nl;
    if (pipeline_ms_total) *pipeline_ms_total = pm;
    if (scan_bytes_per_launch) *scan_bytes_per_launch = nb;
    return VF_OK;
}

// Makespan of the timed main scans: first launch's begin -> last launch's end, on the streams the scans run on.  With
// ordered scans it is the sum of the brackets plus the gaps between launches; with overlapping scans (small shards) it is
// the only well-defined "time per launch": span / launches.  Call after the searches have ended.
extern "C" int vf_index_profile_span(vf_index* ix, double* span_ms, int64_t* launches) {
    if (!ix || !span_ms || !launches) return fail(VF_EINVAL, "vf_index_profile_span: null argument");
    DeviceGuard restore_callers_device;
    vf_index* t = ix->shards.empty() ? ix : ix->shards[0];   // a group: its first shard (equal blocks run in step)
    std::lock_guard<std::mutex> g(t->mu);
    *span_ms = 0.0; *launches = t->prof_launches;
    if (!t->span_started || !t->ev_span) return VF_OK;
    VF_HIP(hipSetDevice(t->device));
    for (int i = 0; i < kSlots; ++i) {
        if (!t->span_slot[i] || !t->slots[i].ev_t[1] || t->slots[i].pending) continue;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, t->ev_span, t->slots[i].ev_t[1]) == hipSuccess && ms > *span_ms) *span_ms = ms;
        else (void)hipGetLastError();
    }
    return VF_OK;
}

// debug: copy the wall-clock stamps of slot's last main scan (option debug bit 7) to host
extern "C" int vf_index_debug_read(vf_index* ix, int32_t slot, unsigned long long* out, int64_t n_words) {
    DeviceGuard restore_callers_device;
    if (!ix || !out || slot < 0 || slot >= kSlots) return fail(VF_EINVAL, "vf_index_debug_read: bad argument");
    if (!ix->shards.empty()) return vf_index_debug_read(ix->shards[0], slot, out, n_words);
    std::lock_guard<std::mutex> g(ix->mu);
    Slot& s = ix->slots[slot];
    const size_t bytes = std::min<size_t>(s.dbg.bytes, (size_t)n_words * 8);
    VF_HIP(hipSetDevice(ix->device));
    VF_HIP(hipDeviceSynchronize());
    if (bytes) VF_HIP(hipMemcpy(out, s.dbg.p, bytes, hipMemcpyDeviceToHost));
    return (int)(bytes / 8);
}

extern "C" int vf_index_slots(vf_index* ix, int32_t* out) {
    if (!ix || !out) return fail(VF_EINVAL, "vf_index_slots: null argument");
    *out = kSlots;
    return VF_OK;
}

extern "C" int vf_index_set_option(vf_index* ix, const char* name, int64_t value) {
    if (!ix || !name) return fail(VF_EINVAL, "vf_index_set_option: null argument");
    for (vf_index* sh : ix->shards) VF_TRY(vf_index_set_option(sh, name, value));  // a group forwards to every shard
    std::lock_guard<std::mutex> g(ix->mu);
    const std::string s(name);
    auto in_range = [&](int64_t lo, int64_t hi) { return value >= lo && value <= hi; };
    if (s == "force_path") { if (!in_range(-1, 2)) return fail(VF_EINVAL, "force_path must be -1 (auto), 0, 1 or 2"); ix->force_path = value; }
    else if (s == "sample_rows") { if (!in_range(-1, 64) || value == 0) return fail(VF_EINVAL, "sample_rows must be -1 (auto) or in [1, 64]"); ix->sample_rows = value; }
    else if (s == "margin") { if (!in_range(-1, 2048)) return fail(VF_EINVAL, "margin must be -1 (auto) or in [0, 2048]"); ix->margin = value; }
    else if (s == "cap") { if (!in_range(0, 16384)) return fail(VF_EINVAL, "cap must be in [0, 16384]"); ix->cap_opt = value; }
    else if (s == "waves") { if (!in_range(0, 8 * 1024)) return fail(VF_EINVAL, "waves must be in [0, 8192]"); ix->waves_opt = value; }
    else if (s == "scan_g") { if (!in_range(0, 4)) return fail(VF_EINVAL, "scan_g must be in [0, 4]"); ix->scan_g = value; }
    else if (s == "refresh_every") { if (!in_range(1, 256)) return fail(VF_EINVAL, "refresh_every must be in [1, 256]"); ix->refresh_every = value; }
    else if (s == "steal") { if (!in_range(0, 1)) return fail(VF_EINVAL, "steal must be 0 or 1"); ix->steal_opt = value; }
    else if (s == "wide") { if (!in_range(0, 4096)) return fail(VF_EINVAL, "wide must be 0 (off), 1 (auto) or a query count"); ix->wide_opt = value; }
    else if (s == "wide8_stage") { if (value != 0 && !in_range(64, 3584)) return fail(VF_EINVAL, "wide8_stage must be 0 (auto) or 64..3584 candidate-stage entries"); ix->wide8_stage = value; }
    else if (s == "wide8_waves") {
#ifdef VF_EXPERIMENTS
        if (value != 4 && value != 8) return fail(VF_EINVAL, "wide8_waves must be 8 (one 512-thread workgroup per CU, 256-query tiles: the default) or 4 (two 256-thread workgroups, 128-query tiles: measured slower, DESIGN.md 4)");
#else
        if (value != 8) return fail(VF_EINVAL, "wide8_waves must be 8 (the 4-wave form was measured slower and is built with -DVF_EXPERIMENTS only: DESIGN.md 4)");
#endif
        ix->wide8_waves = value;
    }
    else if (s == "wide_mfma") { if (!in_range(-1, 1)) return fail(VF_EINVAL, "wide_mfma must be -1 (auto: the fp8 instruction for e4m3 rows), 0 (fp16 matrix instruction on converted rows) or 1 (the fp8 instruction on the e4m3 row bytes: k_scan_wide8)"); ix->wide_mfma = value; }
    else if (s == "wide_sync") { if (!in_range(-1, 8)) return fail(VF_EINVAL, "wide_sync must be -1 (off) or a slack of 0..8 super-tiles"); ix->wide_sync = value; }
    else if (s == "aux_cus") {   // takes effect for slots created afterwards (set it before the first search)
        if (!in_range(-1, 128)) return fail(VF_EINVAL, "aux_cus must be -1 (auto) or in [0, 128]");
        if (ix->aux_applied >= 0 && value != ix->aux_cus)
            return fail(VF_EINVAL, "aux_cus is fixed once the first search has created the scan streams: set it before searching");
        ix->aux_cus = value;
    }
    else if (s == "sample_grid") { if (!in_range(-1, 1024)) return fail(VF_EINVAL, "sample_grid must be -1 (auto), 0 (one workgroup per range) or a workgroup count"); ix->sample_grid = value; }
    else if (s == "scan_impl") { if (!in_range(1, 5)) return fail(VF_EINVAL, "scan_impl must be 1 (k_scan), 2 (auto: k_scan2 for fp16 rows, k_scan2r where it measured faster), 3 (k_scan2 wherever it fits, e4m3 rows converted), 4 (k_scan2 for fp16 rows, never k_scan2r) or 5 (k_scan2r wherever a shape of it exists: fp16 rows of 384 / 512 / 768 / 1024 elements, e4m3 rows of 768 / 1024)"); ix->scan_impl = value; }
    else if (s == "sample_impl") { if (!in_range(-1, 1)) return fail(VF_EINVAL, "sample_impl must be -1 (auto: k_scan2r's operand path for the sample pass where it exists and the CU split is on), 0 (k_scan) or 1 (k_scan2r wherever it fits)"); ix->sample_impl = value; }
    else if (s == "overlap_scans") { if (!in_range(-1, 1)) return fail(VF_EINVAL, "overlap_scans must be -1 (auto), 0 or 1"); ix->overlap_scans = value; }
    else if (s == "debug") ix->debug = value;
    else if (s == "profile") {
        ix->profile = value != 0; ix->prof_scan_ms = ix->prof_pipe_ms = 0.0; ix->prof_launches = 0;
        ix->span_started = false;
        for (bool& b : ix->span_slot) b = false;
    }
    else return fail(VF_EINVAL, "vf_index_set_option: unknown option " + s);
    return VF_OK;
}

// ------------------------------------------------------------------------------------------------
// exact path: canonical dense scores chunk by chunk + LDS sort + running merge.
// qn_dev: [nq][d] canonical normalised queries on device.  Writes [nq][k] at out_ids/out_scores.
// ------------------------------------------------------------------------------------------------
static int exact_search(vf_index* ix, Slot& s, const float* qn_dev, int nq, int k, int64_t* out_ids,
                        float* out_scores, hipStream_t st) {
    const int dt = ix->dtype;
    const int64_t chunk = kSmallN;
    const int64_t nchunks = ix->n == 0 ? 1 : (ix->n + chunk - 1) / chunk;
    if (nchunks > 1) {  // k_merge_topk sorts next_pow2(2k) 8-byte keys in LDS
        int64_t P = 1;
        while (P < 2 * (int64_t)k) P <<= 1;
        if (P * 8 > 160 * 1024) return fail(VF_EUNSUPPORTED, "exact chunked search supports k <= 8192 when n > 16384");
    }
    VF_TRY(s.dense_s.ensure((size_t)nq * chunk * sizeof(float)));
    if (nchunks == 1) {
        const float* cn = ix->cn_cache;
        if (!cn && ix->n > 0) {
            VF_TRY(s.cn_tmp.ensure((size_t)chunk * ix->d * sizeof(float)));
            VF_HIP(launch_normalize_rows(ix->rows_orig, dt, 0, ix->n, ix->d, ix->norm, s.cn_tmp.as<float>(), st));
            cn = s.cn_tmp.as<float>();
        }
        VF_HIP(launch_dense_dot16(qn_dev, nq, cn, ix->n, ix->d, s.dense_s.as<float>(), chunk, st));
        VF_HIP(launch_sort_rows(s.dense_s.as<float>(), chunk, nq, (int)ix->n, k, ix->id_offset, (long long*)out_ids,
                                out_scores, k, st));
        return VF_OK;
    }
    VF_TRY(s.cn_tmp.ensure((size_t)chunk * ix->d * sizeof(float)));
    const size_t part = (size_t)nq * k;
    VF_TRY(s.parts_ids.ensure(2 * part * sizeof(long long)));
    VF_TRY(s.parts_sc.ensure(2 * part * sizeof(float)));
    VF_TRY(s.run_ids.ensure(part * sizeof(long long)));
    VF_TRY(s.run_sc.ensure(part * sizeof(float)));
    long long* pid = s.parts_ids.as<long long>();
    float* psc = s.parts_sc.as<float>();
    VF_HIP(hipMemsetAsync(pid, 0xFF, part * sizeof(long long), st));  // running part = all -1
    VF_HIP(hipMemsetAsync(psc, 0, part * sizeof(float), st));
    for (int64_t c = 0; c < nchunks; ++c) {
        const int64_t r0 = c * chunk, nr = std::min(chunk, ix->n - r0);
        VF_HIP(launch_normalize_rows(ix->rows_orig, dt, r0, nr, ix->d, ix->norm, s.cn_tmp.as<float>(), st));
        VF_HIP(launch_dense_dot16(qn_dev, nq, s.cn_tmp.as<float>(), nr, ix->d, s.dense_s.as<float>(), chunk, st));
        VF_HIP(launch_sort_rows(s.dense_s.as<float>(), chunk, nq, (int)nr, k, ix->id_offset + r0, pid + part, psc + part,
                                k, st));
        VF_HIP(launch_merge_topk(pid, psc, 2, nq, k, s.run_ids.as<long long>(), s.run_sc.as<float>(), st));
        VF_HIP(hipMemcpyAsync(pid, s.run_ids.p, part * sizeof(long long), hipMemcpyDeviceToDevice, st));
        VF_HIP(hipMemcpyAsync(psc, s.run_sc.p, part * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
    VF_HIP(hipMemcpyAsync(out_ids, s.run_ids.p, part * sizeof(long long), hipMemcpyDeviceToDevice, st));
    VF_HIP(hipMemcpyAsync(out_scores, s.run_sc.p, part * sizeof(float), hipMemcpyDeviceToDevice, st));
    return VF_OK;
}

// ------------------------------------------------------------------------------------------------
// path selection + fused pipeline
// ------------------------------------------------------------------------------------------------
struct FusedPlan {
    int kprime, cap, total_waves, grid, samp;
    float eps;
};

static int qn_tile_for(int nq_batch) { return nq_batch <= kQueryTile ? kQueryTile : kMaxBatch; }

static bool fused_possible(const vf_index* ix, int k) {
    if (ix->n <= 1024 || k > kMaxKFused || k <= 0) return false;
    // corpora of up to kSmallN rows are built WITHOUT the scan copy and its inverse norms (they never take the fused path on their
    // own): forcing path 1 on one must be refused, not run on null operands (round 4: found by the option fuzz -- a memory fault)
    if (!ix->rows_scan || !ix->inv_scan) return false;
    if (scan_lds_bytes(ix->dp, kQueryTile) > 160 * 1024) return false;
    return true;
}

static int batch_limit(const vf_index* ix) {
    // 64 queries need dp * 64 * 2 bytes of LDS; fall back to 32-query passes for wide rows
    return scan_lds_bytes(ix->dp, kMaxBatch) <= 160 * 1024 ? kMaxBatch : kQueryTile;
}

static FusedPlan make_plan(const vf_index* ix, int k) {
    FusedPlan p;
    // k' = k + margin, rounded up to a multiple of 32 (whole re-score rounds of 32 row groups)
    const int margin = ix->margin >= 0 ? (int)ix->margin : std::max(24, k / 4);
    p.kprime = ix->margin >= 0 ? k + margin : (k + margin + 31) / 32 * 32;
    p.kprime = std::min(p.kprime, 4096);  // k_final ranks into a fixed 4096-entry LDS array (k <= kMaxKFused = 2048)
    int cap = kMaxCap;
    while (cap < 4 * p.kprime && cap < 16384) cap <<= 1;
    if (ix->cap_opt > 0) { cap = 1; while (cap < ix->cap_opt) cap <<= 1; cap = std::min(cap, 16384); }
    while (cap < 2 * p.kprime) cap <<= 1;
    p.cap = cap;
    const int64_t scan_cus = ix->n_cu - resolved_aux(ix);
    int64_t wgs = std::min<int64_t>(scan_cus, std::max<int64_t>(1, ix->n / 512));
    if (ix->waves_opt > 0) wgs = std::max<int64_t>(1, ix->waves_opt / (kScanThreads / 64));
    p.grid = (int)wgs;
    p.total_waves = p.grid * (kScanThreads / 64);
    // sample rows per wave of the sample pass.  Auto: 16, but 4 for shards of up to 1.1M rows -- there a batch's own chain (k_final of
    // the slot's previous batch -> host turn-around -> prep -> sample pass -> seed -> main scan; two slots in flight) is longer than
    // two scans, so a shorter sample pass shortens the step although the looser seed admits 1.7 x the candidates: configs[1]
    // (1M x 768) 0.304 -> 0.290 ms per batch; from 1.25M rows on the step is the scan's and nothing changes, at 10M the larger
    // candidate lists cost 1.7 % (profiles/r04_sample_rows_sweep.log)
    // (only while the sample still holds 16 k' rows: a top-2048 search seeds its threshold from the k'-th best sample score)
    // Round 6, one box, fresh index per setting (profiles/r06_small_sweep_*.log): 1M rows 4 / 8 / 16 per wave = 0.2965 / 0.2915-0.2951 /
    // 0.3111 ms per batch, 1.25M rows 0.3559 / 0.3515-0.3534 / 0.3539-0.3550, 1.25M x 1024 0.4447 / 0.4479 / 0.4470: 8 is level with the best
    // of the other two at every small-shard size, so it is the rule up to 1.5M rows (16 beyond: the scan hides the pass there).
    p.samp = ix->sample_rows > 0 ? (int)ix->sample_rows : ((ix->n <= 1500000 && 8ll * p.total_waves >= 16ll * p.kprime) ? 8 : 16);
    // A query's candidate list holds about k' (1 + ln(n / sample rows)) entries -- the k'-th best of a growing prefix moves up like that --
    // times the lag of the threshold refresh (measured 1.2-1.3 at k = 100 .. 2048).  The 4 k' rule above is short of that for deep
    // searches over large shards: round 6 found the reference's own call shape, k = 2048 with one to four queries
    // (src/utils/ensembleRetriever.py:64-66), overflowing its 16384-entry lists from 1M rows up and k = 1000 its 8192 -- correct results
    // through the exact re-run, at 56-72 ms instead of 2 (5M rows).  The list is sized for 1.6 x the expectation, up to 32768 entries
    // (what the wide passes use; k_final reads the list from global memory, so its length costs HBM, not LDS).
    if (ix->cap_opt <= 0) {
        const double sample_rows = (double)p.total_waves * p.samp;
        const double expect = p.kprime * (1.0 + log(std::max(1.0, (double)ix->n / std::max(1.0, sample_rows))));
        while (p.cap < (int)(1.6 * expect) && p.cap < 32768) p.cap <<= 1;
    }
    // |approx - canonical| bound (DESIGN.md "Exactness certificate").  fp16 has an 11-bit significand, so
    // round-to-nearest moves an element by at most 2^-11 of its magnitude: rounding the normalised query moves the
    // dot product by <= 2^-11 * sum|q_j c_j| <= 2^-11 (Cauchy-Schwarz, both vectors of unit norm); rounding an fp32
    // corpus row to fp16 adds the same again.  Then the fp16 subnormal floor (2^-25 per element against a unit
    // vector: sqrt(d) * 2^-24 covers it twice) and the two fp32 dot products (d * 2^-24 each).
    const double u16 = 1.0 / 2048.0;
    p.eps = (float)(u16 * (ix->dtype == VF_DTYPE_F32 ? 2.0 : 1.0) + sqrt((double)ix->d) * ldexp(1.0, -24) +
                    2.0 * ix->d * ldexp(1.0, -24) + 1e-6);
    return p;
}

static int select_path(const vf_index* ix, int k) {
    if (ix->force_path >= 0) {
        if (ix->force_path == 1 && !fused_possible(ix, k)) return -1;
        return (int)ix->force_path;
    }
    if (ix->n <= kSmallN) return 0;
    return fused_possible(ix, k) ? 1 : 2;
}

// ---- wide passes (k_scan_wide): up to 1024 queries share ONE read of the shard --------------------------------------
constexpr int kWideMinQueries = 129;   // e4m3 rows: below this the 64-query HBM-bound passes are faster (2 of them at most)
constexpr int kWideMinQueries16 = 65;  // fp16 (and fp32 -> fp16 scan copy) rows: TWO 64-query passes cost two reads of the shard (5.2 ms at 10M x 768), one wide pass 4.2-4.3 ms (round 4, profiles/r04_wide_threshold.log)
constexpr int kWideMaxQueries = 1024;  // 4 query tiles of 256 per pass: one workgroup per CU
constexpr int kWideTile = 256;

static bool wide_possible(const vf_index* ix, int nq) {
    if (ix->wide_opt == 0 || nq < (ix->wide_opt > 1 ? (int)ix->wide_opt : (ix->dtype == VF_DTYPE_FP8_E4M3 ? kWideMinQueries : kWideMinQueries16))) return false;
    // a register stage is 2 k-chunks of fp8 rows / 1 of fp16 rows and a tile alternates two stages
    return ix->dp % (ix->dtype == VF_DTYPE_FP8_E4M3 ? 256 : 128) == 0;
}

static int wide_pass(vf_index* ix, Slot& s, const FusedPlan& p0, const float* d_queries, int nb, int k, int64_t* d_ids,
                     float* d_scores, int flag_off, float* qn_b, bool timed, hipStream_t st, int slot_id) {
    const int qtot = (nb + kWideTile - 1) / kWideTile * kWideTile;
    const int J = qtot / kWideTile;
    const int RG = std::max(1, ix->n_cu / J);
    FusedPlan p = p0;
    if (p.kprime > 256) p.cap = std::max(p.cap, 16384);   // k ~ 1000: ~k' (1 + ln(n / sample)) candidates per query
    // k_scan_wide8 (the fp8 matrix instruction): e4m3 rows, K-tiles of 64, a row group's bytes within a 32-bit lane offset
    const bool w8 = ix->wide_mfma != 0 && ix->dtype == VF_DTYPE_FP8_E4M3 && ix->dp % 64 == 0 &&
                    (ix->n / RG + 2 * 256) * (int64_t)ix->dp < (int64_t)0xFFFFFFFFll;
    if (w8) {
        // the query's hi + lo split leaves ||delta|| ~ 6e-4 of the query's norm (eps_q ~ 1.1e-3 at dp = 1024 against the fp16 path's
        // 6.1e-4): the plan's k' = k + k / 4 still clears it on ordinary data (the k -> k' gap is ~2.4e-3); a deeper k' (k + k / 2) was the
        // first setting and cost 14 % more candidates for nothing (profiles/r04_wide8_kprime.log)
        if (p.kprime > 256) p.cap = std::max(p.cap, 32768);
    }
    const int samp = J >= 2 ? 64 : 32;   // sample rows per row group = samp * 8: 32768 / 65536 rows in all
    const size_t slen = (size_t)RG * samp * 8;
    VF_TRY(s.qimg.ensure((size_t)ix->dp * qtot * 2));
    VF_TRY(s.s0.ensure((size_t)qtot * slen * sizeof(float)));
    VF_TRY(s.cnt.ensure((size_t)qtot * kCntStride * sizeof(u32)));
    VF_TRY(s.tau.ensure((size_t)qtot * sizeof(int)));
    VF_TRY(s.hist.ensure((size_t)qtot * kHistBins * sizeof(u32)));
    VF_TRY(s.hist_coarse.ensure((size_t)qtot * 64 * sizeof(u32)));
    VF_TRY(s.cand.ensure((size_t)qtot * p.cap * sizeof(u64)));
    if (s.wgbase_n != ix->n || s.wgbase_grid != -RG) {   // negative grid tag: the row-group table of the wide scan
        std::vector<long long> base(RG);
        for (int w = 0; w < RG; ++w) base[w] = ix->n * (long long)w / RG;
        VF_TRY(s.wgbase.ensure((size_t)RG * sizeof(long long)));
        VF_HIP(hipMemcpyAsync(s.wgbase.p, base.data(), (size_t)RG * sizeof(long long), hipMemcpyHostToDevice, st));
        VF_HIP(hipStreamSynchronize(st));
        s.wgbase_n = ix->n; s.wgbase_grid = -RG;
    }
    VF_HIP(launch_prep_queries(d_queries, nb, ix->d, ix->dp, qtot, qn_b, s.qimg.as<_Float16>(), st));
    if (w8) {
        VF_TRY(s.qimg8.ensure((size_t)ix->dp * qtot * 2));
        VF_TRY(s.epsq.ensure((size_t)qtot * sizeof(float)));
        VF_HIP(launch_prep_wide8(qn_b, nb, ix->d, ix->dp, qtot, (unsigned char*)s.qimg8.p, s.epsq.as<float>(), st));
    }
    const bool f8 = ix->dtype == VF_DTYPE_FP8_E4M3;
    ScanArgs a{};
    a.rows = (const char*)ix->rows_scan; a.inv_scan = ix->inv_scan; a.qimg = s.qimg.as<_Float16>();
    a.n = ix->n; a.dp = ix->dp; a.row_bytes = (long long)ix->dp * (f8 ? 1 : 2);
    a.total_waves = RG * 8; a.samp = samp;
    a.s0 = s.s0.as<float>(); a.wg_base = s.wgbase.as<long long>(); a.cnt = s.cnt.as<u32>(); a.tau_bin = s.tau.as<int>();
    a.hist = s.hist.as<u32>(); a.hist_coarse = s.hist_coarse.as<u32>(); a.cand = s.cand.as<u64>(); a.cap = p.cap; a.kprime = p.kprime;
    a.stage_cap = 3584; a.dbg = nullptr; a.debug = (int)ix->debug;   // 56 KB of candidate stage: the LDS the 96 KB of operand buffers and the control block leave
    a.refresh_every = 1;   // a power of two on this path (block completion is tested with a mask)
    while (a.refresh_every * 2 <= (int)std::min<int64_t>(256, std::max<int64_t>(1, ix->refresh_every))) a.refresh_every *= 2;
    a.nq = nb; a.qn_total = qtot; a.jtiles = J; a.rgroups = RG;
    if (ix->n / RG < (int64_t)samp * 8)   // sample slots no wave writes must read as empty (NaN)
        VF_HIP(hipMemsetAsync(a.s0, 0xFF, (size_t)qtot * slen * sizeof(float), st));
    VF_HIP(launch_scan_wide(a, kModeSample, f8, st));
    VF_HIP(launch_sel0(a, qtot, st));
    for (int o = 0; o < kSlots; ++o)
        if (o != slot_id && ix->slots[o].ev_scan) VF_HIP(hipStreamWaitEvent(st, ix->slots[o].ev_scan, 0));
    const int w8_waves = ix->wide8_waves == 4 ? 4 : 8;
    const int J8 = w8 && w8_waves == 4 ? (nb + 127) / 128 : J;   // query tiles of the main pass: 128 wide for the two-workgroups-per-CU form
    if (J8 > 1 && ix->wide_sync >= 0) {   // sibling progress words of the main pass (see k_scan_wide; eight per row group)
        VF_TRY(s.sib.ensure((size_t)RG * 8 * sizeof(u32)));
        VF_HIP(hipMemsetAsync(s.sib.p, 0, (size_t)RG * 8 * sizeof(u32), st));
        a.sib = s.sib.as<u32>(); a.sib_slack = (int)ix->wide_sync;
    }
    if (timed) VF_HIP(hipEventRecord(s.ev_t[0], st));
    if (w8) {
        ScanArgs a8 = a;
        a8.qimg = (const _Float16*)s.qimg8.p;
        a8.jtiles = J8;
        a8.stage_cap = ix->wide8_stage > 0 ? (int)std::min<int64_t>(ix->wide8_stage, scan_wide8_stage_cap(w8_waves)) : scan_wide8_stage_cap(w8_waves);
        if (ix->debug & 128) {   // per-wave phase times of k_scan_wide8 (vf_index_debug_read)
            VF_TRY(s.dbg.ensure((size_t)8 * J8 * ((RG + 7) / 8) * w8_waves * 16 * sizeof(u64)));
            VF_HIP(hipMemsetAsync(s.dbg.p, 0, s.dbg.bytes, st));
            a8.dbg = s.dbg.as<u64>();
        }
        VF_HIP(launch_scan_wide8(a8, w8_waves, st));
    } else
        VF_HIP(launch_scan_wide(a, kModeMain, f8, st));
    VF_HIP(hipEventRecord(s.ev_scan, st));
    if (timed) {
        VF_HIP(hipEventRecord(s.ev_t[1], st));
        const int64_t sampled = std::min<int64_t>(ix->n, (int64_t)RG * std::min<int64_t>((int64_t)samp * 8, ix->n / RG));
        ix->prof_bytes = (ix->n - sampled) * ((int64_t)ix->d * (f8 ? 1 : 2) + 4);
    }
    FinalArgs f{};
    f.cnt = a.cnt; f.cand = a.cand; f.cap = p.cap; f.tau_bin = a.tau_bin; f.rows_orig = ix->rows_orig;
    f.orig_dtype = ix->dtype; f.orig_row_elems = ix->d; f.norm = ix->norm; f.qn = qn_b;
    f.d = ix->d; f.k = k; f.kprime = p.kprime; f.eps = p.eps; f.n_rows = ix->n; f.id_offset = ix->id_offset;
    f.eps_q = w8 ? s.epsq.as<float>() : nullptr;   // (the sample pass scored with fp16 queries: its bound is the smaller one)
    f.out_ids = (long long*)d_ids; f.out_scores = d_scores;
    f.flags = s.d_flags + flag_off; f.cand_count_out = s.d_counts + flag_off;
    f.dbg = nullptr;
    if (ix->debug & 256) { VF_TRY(s.dbg.ensure((size_t)qtot * 8 * sizeof(u64))); f.dbg = s.dbg.as<u64>(); }
    VF_HIP(launch_final(f, nb, st));
    s.scan_kernel = w8 ? 4 : 3;
    return VF_OK;
}

static int begin_impl(vf_index* ix, int slot_id, const float* d_queries, int nq, int k, int64_t* d_ids,
                      float* d_scores, hipStream_t user) {
    Slot& s = ix->slots[slot_id];
    if (s.pending) return fail(VF_EINVAL, "vf_index_search_begin: slot already has a pending search");
    const int path = select_path(ix, k);
    if (path < 0) return fail(VF_EUNSUPPORTED, "forced fused path is not possible for this n / k / d");
    if (path != 1 && ix->n > kSmallN && k > 8192)  // checked before anything is enqueued (k_merge_topk's LDS sort)
        return fail(VF_EUNSUPPORTED, "exact chunked search supports k <= 8192 when n > 16384");
    VF_TRY(ensure_slot(ix, s));
    VF_HIP(hipEventRecord(s.ev_in, user));
    VF_HIP(hipStreamWaitEvent(s.stream, s.ev_in, 0));
    hipStream_t st = s.stream;
    const int bl = batch_limit(ix);
    VF_TRY(s.qn.ensure((size_t)std::max(nq, 1) * ix->d * sizeof(float)));
    VF_TRY(s.qimg.ensure(scan_lds_bytes(ix->dp, kMaxBatch)));
    VF_TRY(ensure_pinned(s, (size_t)nq));
    s.pending = true; s.d_queries = d_queries; s.nq = nq; s.k = k; s.d_ids = d_ids; s.d_scores = d_scores;
    s.path = path; s.user_stream = user; s.wide_launches = 0; s.wide_queries = 0; s.scan_kernel = 0;
    if (nq == 0 || k == 0) { VF_HIP(hipEventRecord(s.ev_done, st)); return VF_OK; }

    if (path != 1) {
        for (int b0 = 0; b0 < nq; b0 += bl) {
            const int nb = std::min(bl, nq - b0);
            VF_HIP(launch_prep_queries(d_queries + (size_t)b0 * ix->d, nb, ix->d, ix->dp, qn_tile_for(nb),
                                       s.qn.as<float>() + (size_t)b0 * ix->d, s.qimg.as<_Float16>(), st));
            VF_TRY(exact_search(ix, s, s.qn.as<float>() + (size_t)b0 * ix->d, nb, k, d_ids + (size_t)b0 * k,
                                d_scores + (size_t)b0 * k, st));
        }
        VF_HIP(hipEventRecord(s.ev_done, st));
        return VF_OK;
    }

    FusedPlan p = make_plan(ix, k);
    if (wide_possible(ix, nq)) {
        s.timed = ix->profile;
        if (s.timed) VF_HIP(hipEventRecord(s.ev_t[2], st));
        for (int b0 = 0; b0 < nq; b0 += kWideMaxQueries) {
            const int nb = std::min(kWideMaxQueries, nq - b0);
            VF_TRY(wide_pass(ix, s, p, d_queries + (size_t)b0 * ix->d, nb, k, d_ids + (size_t)b0 * k, d_scores + (size_t)b0 * k,
                             b0, s.qn.as<float>() + (size_t)b0 * ix->d, s.timed && b0 == 0, st, slot_id));
            ++s.wide_launches; s.wide_queries += nb;
        }
        if (s.timed) VF_HIP(hipEventRecord(s.ev_t[3], st));
        VF_HIP(hipEventRecord(s.ev_done, st));
        return VF_OK;
    }
    VF_TRY(s.s0.ensure((size_t)kMaxBatch * p.total_waves * p.samp * sizeof(float)));
    VF_TRY(s.cnt.ensure((size_t)kMaxBatch * kCntStride * sizeof(u32)));
    VF_TRY(s.tau.ensure(kMaxBatch * sizeof(int)));
    VF_TRY(s.hist.ensure((size_t)kMaxBatch * kHistBins * sizeof(u32)));
    VF_TRY(s.hist_coarse.ensure((size_t)kMaxBatch * 64 * sizeof(u32)));
    VF_TRY(s.cand.ensure((size_t)kMaxBatch * p.cap * sizeof(u64)));
    VF_TRY(s.tilecnt.ensure((size_t)p.grid * sizeof(u32)));
    s.timed = ix->profile;
    if (s.timed) VF_HIP(hipEventRecord(s.ev_t[2], st));
    if (s.wgbase_n != ix->n || s.wgbase_grid != p.grid) {  // first row of every scan workgroup (k_sel0 maps sample slots back to rows)
        std::vector<long long> base(p.grid);
        for (int w = 0; w < p.grid; ++w) base[w] = ix->n * (long long)w / p.grid;
        VF_TRY(s.wgbase.ensure((size_t)p.grid * sizeof(long long)));
        VF_HIP(hipMemcpy(s.wgbase.p, base.data(), (size_t)p.grid * sizeof(long long), hipMemcpyHostToDevice));
        s.wgbase_n = ix->n; s.wgbase_grid = p.grid;
    }
    for (int b0 = 0; b0 < nq; b0 += bl) {
        const int nb = std::min(bl, nq - b0);
        const int qt = qn_tile_for(nb);
        float* qn_b = s.qn.as<float>() + (size_t)b0 * ix->d;
        VF_HIP(launch_prep_queries(d_queries + (size_t)b0 * ix->d, nb, ix->d, ix->dp, qt, qn_b, s.qimg.as<_Float16>(), st));
        ScanArgs a{};
        a.rows = (const char*)ix->rows_scan; a.inv_scan = ix->inv_scan; a.qimg = s.qimg.as<_Float16>();
        a.n = ix->n; a.dp = ix->dp; a.row_bytes = (long long)ix->dp * (ix->dtype == VF_DTYPE_FP8_E4M3 ? 1 : 2); a.total_waves = p.total_waves; a.samp = p.samp;
        a.s0 = s.s0.as<float>(); a.wg_base = s.wgbase.as<long long>(); a.cnt = s.cnt.as<u32>(); a.tau_bin = s.tau.as<int>(); a.hist = s.hist.as<u32>();
        a.cand = s.cand.as<u64>(); a.cap = p.cap; a.kprime = p.kprime;
        a.hist_coarse = s.hist_coarse.as<u32>(); a.stage_cap = scan_stage_cap(ix->dp, qt);
        a.tile_cnt = ix->steal_opt ? s.tilecnt.as<u32>() : nullptr; a.scan_grid = p.grid;
        a.dbg = nullptr;
        if (ix->debug & 128) { VF_TRY(s.dbg.ensure((size_t)p.total_waves * ((ix->debug & 512) ? 72 : 4) * sizeof(u64))); a.dbg = s.dbg.as<u64>(); if (ix->debug & 512) VF_HIP(hipMemsetAsync(s.dbg.p, 0, s.dbg.bytes, st)); }
        // A workgroup publishes its staged candidates (and refreshes one threshold) per BLOCK of refresh_every staged entries,
        // whatever query they belong to: the option is stated for a full 64-query batch and scales with the batch's query
        // count, so that a query sees the same publication granularity at nq = 1 as at nq = 64.  (Unscaled, a single
        // query over 5M rows staged ~76 entries per workgroup, never completed a 128-entry block, never raised its
        // threshold above the sample's and overflowed its candidate list: exact re-run, 38 ms instead of 1.5.)
        // (rounded down to a power of two: the kernels find a block's number with a shift)
        a.refresh_every = (int)std::min<int64_t>(256, std::max<int64_t>(4, std::max<int64_t>(1, ix->refresh_every) * nb / kMaxBatch));
        while (a.refresh_every & (a.refresh_every - 1)) a.refresh_every &= a.refresh_every - 1;
        a.nq = nb; a.debug = (int)ix->debug;
        // sample slots no wave writes (a wave range shorter than samp) must read as empty: 0xFF bytes
        // are a NaN, which k_sel0's "v > -inf" test skips.  Never needed once n >= TW * samp.
        if (ix->n / p.total_waves < p.samp)
            VF_HIP(hipMemsetAsync(a.s0, 0xFF, (size_t)qt * p.total_waves * p.samp * sizeof(float), st));
        // sample pass: a FEW workgroups walk the sample parts of all ranges (each stages the query image once)
        // (auto: 4 workgroups per spare CU when the CU split is on, one per range otherwise)
        // Round 6: where k_scan2r's operand path is the default (scan2r_auto_width) the sample pass takes it too -- ONE workgroup per spare
        // CU, each walking the sample parts of p.grid / 32 ranges with six-segment rings: the pass is bound by what a CU keeps in flight
        // (k_scan's register-staged loads: 68-71 us for 8 rows per wave in four rounds of 128 workgroups).  sample_impl: -1 auto, 0 k_scan, 1 k_scan2r
        const bool f8rows = ix->dtype == VF_DTYPE_FP8_E4M3;
        // e4m3 rows (768 / 1024 elements) take it wherever k_scan2r is their main scan (n > 1.1M: below), whole chip or split.
        const bool r_f8_auto = f8rows && ix->scan_impl == 2 && ix->n > kScan2rMinRows && !ix->steal_opt && scan2r_auto_width(ix->dp, true);
        const bool sample_r = ix->sample_impl != 0 && ix->scan_impl != 1 && scan2r_stage_cap(ix->dp, qt, f8rows) >= 256 &&
                              (ix->sample_impl == 1 || (!f8rows && s.scan_stream != s.stream && scan2r_auto_width(ix->dp, false)) || r_f8_auto);
        if (sample_r) {
            const int64_t sg_r = ix->sample_grid > 0 ? ix->sample_grid : (s.scan_stream != s.stream ? resolved_aux(ix) : p.grid);
            ScanArgs as = a;
            as.stage_cap = 0;
            VF_HIP(launch_scan2r_sample(as, qt, (int)std::min<int64_t>(std::max<int64_t>(sg_r, 1), p.grid), f8rows, st));
        } else {
        const int64_t sg_opt = ix->sample_grid >= 0 ? ix->sample_grid : (s.scan_stream != s.stream ? 4 * resolved_aux(ix) : 0);
        const int sgrid = sg_opt > 0 ? (int)std::min<int64_t>(sg_opt, p.grid) : p.grid;
        VF_HIP(launch_scan(a, kModeSample, qt, sgrid, (int)ix->scan_g, ix->dtype == VF_DTYPE_FP8_E4M3, st));
        }
        VF_HIP(launch_sel0(a, qt, st));
        hipStream_t sst = s.scan_stream;
        if (sst != st) {   // the main scan runs on the CU-masked stream, behind this slot's prologue
            VF_HIP(hipEventRecord(s.ev_pro, st));
            VF_HIP(hipStreamWaitEvent(sst, s.ev_pro, 0));
        }
        // Main scans of different slots run one after the other (a scan workgroup owns its CU); ordering them explicitly
        // keeps queueing time out of the timed bracket.  With overlap_scans they are left to the dispatcher: the next
        // scan's workgroups start on the CUs the previous one has finished with (no idle tail), at the price of that bracket.
        if (!resolved_overlap(ix))
            for (int o = 0; o < kSlots; ++o)
                if (o != slot_id && ix->slots[o].ev_scan) VF_HIP(hipStreamWaitEvent(sst, ix->slots[o].ev_scan, 0));
        if (s.timed && b0 == 0) {
            if (!ix->span_started) {   // makespan of the timed launches: from here to the last launch's end (vf_index_profile_span)
                if (!ix->ev_span) VF_HIP(hipEventCreate(&ix->ev_span));
                VF_HIP(hipEventRecord(ix->ev_span, sst));
                ix->span_started = true;
            }
            ix->span_slot[slot_id] = true;
            VF_HIP(hipEventRecord(s.ev_t[0], sst));
        }
        const int f8 = ix->dtype == VF_DTYPE_FP8_E4M3 ? 1 : 0;
        // k_scan2 serves fp16 rows by default; e4m3 rows only on request (scan_impl = 3, or 5 for k_scan2r's e4m3 shapes): per byte they
        // carry twice the matrix work and the same LDS-DMA issues, and with ONE wave per SIMD nothing hides either -- measured 0.53
        // (k_scan2, round 3) and 0.55-0.60 (k_scan2r, round 6: B fragments in accumulator registers, rings of six) against k_scan's
        // 0.63-0.70 of peak at 10M x 768 / 1024 fp8 (profiles/r03_f8_sweep.log, r06_fp8_scan2r_ab.log)
        const int cap2 = ((ix->scan_impl == 3 || ((ix->scan_impl == 2 || ix->scan_impl == 4 || ix->scan_impl == 5) && !f8)) && !ix->steal_opt) ? scan2_stage_cap(ix->dp, qt, f8) : 0;
        // k_scan2r (round 6): part of the query image in accumulator registers, deeper rings.  fp16 rows of 768 elements, measured against
        // k_scan2 in separate processes, alternating (profiles/r06_scan2r_ab.log): the 8-GPU rank's shard (1.25M rows) 0.3469-0.3528 ms
        // per batch against 0.3538-0.3602 (2.2 % faster: a wave keeps 24 KB in flight instead of 12), 10M rows level (2.538 vs 2.548 --
        // the scan sits on the copy ceiling there), configs[1] (1M rows) 3 % SLOWER (0.303-0.315 vs 0.293-0.303: that step is the
        // prologue chain's, and the workgroup's start is 2.3 us longer).  So: auto (scan_impl = 2) takes it above 1.1M rows wherever the
        // scans run on the CU split and overlap (which, for these rows, is every size: split_limit); 5 forces it, 4 forbids it.
        // e4m3 rows (round 6, after the filter rewrite): k_scan2r was 0.55-0.60 against k_scan's 0.63-0.70 while a tile's threshold filter
        // cost a lone wave 4 500 cycles; with the filter at ~1 000 it is 0.694-0.698 against 0.627-0.656 at 10M x 768 and 0.717-0.719
        // against 0.693-0.700 at 10M x 1024 (whole chip, ordered scans; split + overlap loses 3-5 % there), +2-3 % at 1.25M rows with its
        // own sample pass, level at 1M: the same row threshold as fp16 rows, no CU-split condition (profiles/r06_after_filter_kernel_choice.log)
        const bool r_auto = ix->scan_impl == 2 && ix->n > kScan2rMinRows && scan2r_auto_width(ix->dp, f8 != 0) && (f8 || (s.scan_stream != s.stream && resolved_overlap(ix)));   // (fp16 rows: with the CU split and overlapping scans only: above)
#ifdef VF_EXPERIMENTS
        const bool dbg_r = f8 || !(ix->debug & (32 | 64));   // (bits 5 / 6 are k_scan2's experiments on fp16 rows, k_scan2r's on e4m3 rows)
#else
        const bool dbg_r = true;
#endif
        const int capr = ((ix->scan_impl == 5 || r_auto) && !ix->steal_opt && dbg_r) ? scan2r_stage_cap(ix->dp, qt, f8) : 0;
        if (capr >= 256) {
            ScanArgs a2 = a;
            a2.stage_cap = capr;
            VF_HIP(launch_scan2r(a2, qt, p.grid, f8, sst));
            s.scan_kernel = 5;
        } else if (cap2 >= 256) {   // whole-line LDS-DMA loads: image + four rings + a stage of >= 256 entries fit the 160 KB
            ScanArgs a2 = a;
            a2.stage_cap = cap2;
#ifdef VF_EXPERIMENTS
            if (ix->debug & 32) {   // timing experiment: compute unit -> range table (k_scan2, debug bit 5)
                const bool fresh = s.sib.bytes < 4096;
                VF_TRY(s.sib.ensure(4096));
                if (fresh) VF_HIP(hipMemsetAsync(s.sib.p, 0xFF, 4096, st));
                a2.sib = s.sib.as<u32>();
            }
#endif
            VF_HIP(launch_scan2(a2, qt, p.grid, f8, sst));
            s.scan_kernel = 2;
        } else {
            VF_HIP(launch_scan(a, kModeMain, qt, p.grid, (int)ix->scan_g, ix->dtype == VF_DTYPE_FP8_E4M3, sst));
            s.scan_kernel = 1;
        }
        VF_HIP(hipEventRecord(s.ev_scan, sst));
        if (s.timed && b0 == 0) {
            VF_HIP(hipEventRecord(s.ev_t[1], sst));
            const int64_t per_wave = ix->n / p.total_waves;
            const int64_t sampled = std::min<int64_t>(ix->n, (int64_t)p.total_waves * std::min<int64_t>(p.samp, per_wave));
            ix->prof_bytes = (ix->n - sampled) * ((int64_t)ix->d * (ix->dtype == VF_DTYPE_FP8_E4M3 ? 1 : 2) + 4);
        }
        if (sst != st) VF_HIP(hipStreamWaitEvent(st, s.ev_scan, 0));
        FinalArgs f{};
        f.cnt = a.cnt; f.cand = a.cand; f.cap = p.cap; f.tau_bin = a.tau_bin; f.rows_orig = ix->rows_orig;
        f.orig_dtype = ix->dtype; f.orig_row_elems = ix->d; f.norm = ix->norm; f.qn = qn_b;
        f.d = ix->d; f.k = k; f.kprime = p.kprime; f.eps = p.eps; f.n_rows = ix->n; f.id_offset = ix->id_offset;
        f.out_ids = (long long*)(d_ids + (size_t)b0 * k); f.out_scores = d_scores + (size_t)b0 * k;
        f.flags = s.d_flags + b0; f.cand_count_out = s.d_counts + b0;
        f.dbg = nullptr;
        if (ix->debug & 256) { VF_TRY(s.dbg.ensure((size_t)std::max(p.total_waves * 4, 64 * 8) * sizeof(u64))); f.dbg = s.dbg.as<u64>(); }
        VF_HIP(launch_final(f, nb, st));
    }
    if (s.timed) VF_HIP(hipEventRecord(s.ev_t[3], st));
    VF_HIP(hipEventRecord(s.ev_done, st));
    return VF_OK;
}

// (Round 6 measured a poll-then-block wait here -- hipEventQuery for up to 3 ms before hipEventSynchronize -- against the plain blocking
//  call, alternating processes, at 1M / 1.25M / 10M rows: no difference (0.2939-0.2943 vs 0.2933-0.2954 ms per batch at 1M): the
//  runtime's own wait already polls.  profiles/r06_spin_wait_ab.log)
static int end_impl(vf_index* ix, int slot_id) {
    Slot& s = ix->slots[slot_id];
    if (!s.pending) return fail(VF_EINVAL, "vf_index_search_end: slot has no pending search");
    s.pending = false;
    VF_HIP(hipEventSynchronize(s.ev_done));
    vf_search_stats stt{};
    stt.path = s.path; stt.n_queries = s.nq; stt.wide_launches = s.wide_launches; stt.wide_queries = s.wide_queries;
    stt.aux_cus = (s.path == 1 && s.scan_stream && s.scan_stream != s.stream) ? resolved_aux(ix) : 0;
    stt.scans_overlap = (s.path == 1 && resolved_overlap(ix)) ? 1 : 0;
    stt.scan_kernel = s.scan_kernel;
    if (s.path == 1 && s.nq > 0 && s.k > 0) {
        if (s.timed) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, s.ev_t[0], s.ev_t[1]) == hipSuccess) { ix->prof_scan_ms += ms; ++ix->prof_launches; }
            if (hipEventElapsedTime(&ms, s.ev_t[2], s.ev_t[3]) == hipSuccess) ix->prof_pipe_ms += ms;
            s.timed = false;
        }
        std::vector<int> redo;
        for (int q = 0; q < s.nq; ++q) {
            stt.candidates += s.h_counts[q];
            stt.max_candidates = std::max<int64_t>(stt.max_candidates, s.h_counts[q]);
            if (s.h_flags[q] == 1) ++stt.uncertified;
            if (s.h_flags[q] == 2) ++stt.overflowed;
            if (s.h_flags[q] != 0) redo.push_back(q);
        }
        if (!redo.empty()) {
            // repair: exact chunked search for the flagged queries only (still on the GPU)
            const int m = (int)redo.size();
            hipStream_t st = s.stream;
            VF_TRY(s.qsel.ensure((size_t)m * ix->d * sizeof(float) + (size_t)m * s.k * (sizeof(long long) + sizeof(float))));
            float* qsel = s.qsel.as<float>();
            long long* rid = (long long*)(qsel + (size_t)m * ix->d);
            float* rsc = (float*)(rid + (size_t)m * s.k);
            for (int i = 0; i < m; ++i)
                VF_HIP(hipMemcpyAsync(qsel + (size_t)i * ix->d, s.qn.as<float>() + (size_t)redo[i] * ix->d,
                                      (size_t)ix->d * sizeof(float), hipMemcpyDeviceToDevice, st));
            for (int i0 = 0; i0 < m; i0 += kMaxBatch) {
                const int nb = std::min(kMaxBatch, m - i0);
                VF_TRY(exact_search(ix, s, qsel + (size_t)i0 * ix->d, nb, s.k, (int64_t*)(rid + (size_t)i0 * s.k),
                                    rsc + (size_t)i0 * s.k, st));
            }
            for (int i = 0; i < m; ++i) {
                VF_HIP(hipMemcpyAsync(s.d_ids + (size_t)redo[i] * s.k, rid + (size_t)i * s.k, (size_t)s.k * sizeof(long long),
                                      hipMemcpyDeviceToDevice, st));
                VF_HIP(hipMemcpyAsync(s.d_scores + (size_t)redo[i] * s.k, rsc + (size_t)i * s.k, (size_t)s.k * sizeof(float),
                                      hipMemcpyDeviceToDevice, st));
            }
            VF_HIP(hipEventRecord(s.ev_done, st));
            VF_HIP(hipEventSynchronize(s.ev_done));
            stt.exact_reruns = m;
        }
    }
    // later work on the caller's stream sees the results
    VF_HIP(hipStreamWaitEvent(s.user_stream, s.ev_done, 0));
    ix->stats = stt;
    return VF_OK;
}

static int check_search_args(vf_index* ix, const void* q, int nq, int k, const void* ids, const void* sc) {
    if (!ix) return fail(VF_EINVAL, "search: null handle");
    if (nq < 0 || k < 0) return fail(VF_EINVAL, "search: negative nq / k");
    if (nq > 0 && k > 0 && (!q || !ids || !sc)) return fail(VF_EINVAL, "search: null buffer");
    return VF_OK;
}

// ------------------------------------------------------------------------------------------------
// Group handle: ONE process, several devices (the reference's serve path is a single process: RAGManager singleton,
// src/utils/ragManager.py:17-30, building EnsembleRetriever -> FaissRetriever at ensembleRetriever.py:39-43).  Rows are
// split into contiguous blocks, one vf_index per device; a search copies the queries from the home
// device to every shard over xGMI peer copies, runs all shard searches concurrently (each on its own device streams),
// copies the packed per-shard top-k back to the home device and merges them there with k_merge_topk.  No host thread
// per device and no RCCL communicator are needed: the exchange is G small peer-to-peer copies (77 KB each at nq = 64,
// k = 100), which is what an all-gather over point-to-point xGMI links amounts to for one receiver.
// ------------------------------------------------------------------------------------------------
static int group_ensure_slot(vf_index* ix, GroupSlot& gs) {
    const size_t G = ix->shards.size();
    if (gs.feed.size() == G) return VF_OK;
    gs.feed.assign(G, nullptr); gs.ev.assign(G, nullptr);
    gs.dq.resize(G); gs.blob.resize(G);
    for (size_t g = 0; g < G; ++g) {
        VF_HIP(hipSetDevice(ix->shards[g]->device));
        VF_HIP(hipStreamCreateWithFlags(&gs.feed[g], hipStreamNonBlocking));
        VF_HIP(hipEventCreateWithFlags(&gs.ev[g], hipEventDisableTiming));
    }
    VF_HIP(hipSetDevice(ix->device));
    VF_HIP(hipEventCreateWithFlags(&gs.ev_in, hipEventDisableTiming));
    gs.hfeed.assign(G, nullptr); gs.ev_hb.assign(G, nullptr); gs.ev_home.assign(G, nullptr); gs.hb.resize(G); gs.staged_last.assign(G, 0);
    return VF_OK;
}

static bool shard_is_staged(const vf_index* ix, size_t g) { return g < ix->peer_ok.size() && !ix->peer_ok[g]; }

// streams / events of the host-staged exchange for shard g (first use)
static int group_ensure_staged(vf_index* ix, GroupSlot& gs, size_t g) {
    if (!gs.hq_stream) {
        VF_HIP(hipSetDevice(ix->device));
        VF_HIP(hipStreamCreateWithFlags(&gs.hq_stream, hipStreamNonBlocking));
        VF_HIP(hipEventCreateWithFlags(&gs.ev_hq, hipEventDisableTiming));
    }
    if (!gs.hfeed[g]) {
        VF_HIP(hipSetDevice(ix->device));
        VF_HIP(hipStreamCreateWithFlags(&gs.hfeed[g], hipStreamNonBlocking));
        VF_HIP(hipEventCreateWithFlags(&gs.ev_home[g], hipEventDisableTiming));
        VF_HIP(hipSetDevice(ix->shards[g]->device));
        VF_HIP(hipEventCreateWithFlags(&gs.ev_hb[g], hipEventDisableTiming));
    }
    return VF_OK;
}

static int group_begin(vf_index* ix, int slot_id, const float* d_queries, int nq, int k, int64_t* d_ids, float* d_scores,
                       hipStream_t user) {
    GroupSlot& gs = ix->gslots[slot_id];
    if (gs.pending) return fail(VF_EINVAL, "vf_index_search_begin: slot already has a pending search");
    const size_t G = ix->shards.size();
    if ((size_t)G * k > 16384) return fail(VF_EUNSUPPORTED, "sharded search: shards * k > 16384");
    VF_TRY(group_ensure_slot(ix, gs));
    gs.nq = nq; gs.k = k; gs.d_ids = d_ids; gs.d_scores = d_scores; gs.user = user;
    if (nq == 0 || k == 0) { gs.pending = true; return VF_OK; }
    const size_t qbytes = (size_t)nq * ix->d * sizeof(float);
    const size_t part = (size_t)packed_part_bytes(nq, k);
    VF_HIP(hipSetDevice(ix->device));
    VF_TRY(gs.all.ensure(G * part));
    VF_HIP(hipEventRecord(gs.ev_in, user));
    size_t started = 0;
    int rc = VF_OK;
    bool host_copy_issued = false;
    for (size_t g = 0; g < G && rc == VF_OK; ++g) {
        vf_index* sh = ix->shards[g];
        const bool staged = shard_is_staged(ix, g);
        hipError_t e = hipSuccess;
        if (staged) {   // home -> pinned host (once per search, on a home-device stream) -> shard g (on its feed stream)
            if ((rc = group_ensure_staged(ix, gs, g)) != VF_OK) break;
            if (!host_copy_issued) {
                if ((rc = gs.hq.ensure(qbytes)) != VF_OK) break;
                e = hipSetDevice(ix->device);
                if (e == hipSuccess) e = hipStreamWaitEvent(gs.hq_stream, gs.ev_in, 0);
                if (e == hipSuccess) e = hipMemcpyAsync(gs.hq.p, d_queries, qbytes, hipMemcpyDeviceToHost, gs.hq_stream);
                if (e == hipSuccess) e = hipEventRecord(gs.ev_hq, gs.hq_stream);
                if (e != hipSuccess) { rc = fail(VF_EHIP, std::string("sharded search: staged query copy (home -> host): ") + hipGetErrorString(e)); break; }
                host_copy_issued = true;
            }
        }
        e = hipSetDevice(sh->device);
        if (e == hipSuccess) e = hipStreamWaitEvent(gs.feed[g], staged ? gs.ev_hq : gs.ev_in, 0);
        if (e != hipSuccess) { rc = fail(VF_EHIP, std::string("sharded search: ") + hipGetErrorString(e)); break; }
        if ((rc = gs.dq[g].ensure(qbytes)) != VF_OK || (rc = gs.blob[g].ensure(part)) != VF_OK) break;
        if (staged) e = hipMemcpyAsync(gs.dq[g].p, gs.hq.p, qbytes, hipMemcpyHostToDevice, gs.feed[g]);
        else e = hipMemcpyPeerAsync(gs.dq[g].p, sh->device, d_queries, ix->device, qbytes, gs.feed[g]);
        if (e != hipSuccess) { rc = fail(VF_EHIP, std::string("sharded search: query ") + (staged ? "staged" : "peer") + " copy: " + hipGetErrorString(e)); break; }
        std::lock_guard<std::mutex> lk(sh->mu);
        rc = begin_impl(sh, slot_id, gs.dq[g].as<float>(), nq, k, (int64_t*)gs.blob[g].p,
                        (float*)((char*)gs.blob[g].p + (size_t)nq * k * 8), gs.feed[g]);
        if (rc == VF_OK) ++started; else sh->slots[slot_id].pending = false;
    }
    if (rc != VF_OK) {  // let the shards that did start finish, then report
        const std::string keep = g_err;
        for (size_t g = 0; g < started; ++g) {
            (void)hipSetDevice(ix->shards[g]->device);
            std::lock_guard<std::mutex> lk(ix->shards[g]->mu);
            (void)end_impl(ix->shards[g], slot_id);
        }
        g_err = keep;
        return rc;
    }
    gs.pending = true;
    return VF_OK;
}

static int group_end(vf_index* ix, int slot_id) {
    GroupSlot& gs = ix->gslots[slot_id];
    if (!gs.pending) return fail(VF_EINVAL, "vf_index_search_end: slot has no pending search");
    gs.pending = false;
    vf_search_stats tot{};
    tot.n_queries = gs.nq;
    if (gs.nq == 0 || gs.k == 0) { ix->stats = tot; return VF_OK; }
    const size_t G = ix->shards.size();
    const size_t part = (size_t)packed_part_bytes(gs.nq, gs.k);
    int rc = VF_OK;
    for (size_t g = 0; g < G; ++g) {  // every shard is ended even after a failure: no slot may stay pending
        vf_index* sh = ix->shards[g];
        (void)hipSetDevice(sh->device);
        int r;
        {
            std::lock_guard<std::mutex> lk(sh->mu);
            r = end_impl(sh, slot_id);  // waits for the shard, certifies, repairs; feed[g] is ordered after the result
            if (r == VF_OK) {
                const vf_search_stats& st = sh->stats;
                tot.path = std::max(tot.path, st.path);
                tot.candidates += st.candidates; tot.max_candidates = std::max(tot.max_candidates, st.max_candidates);
                tot.uncertified += st.uncertified; tot.overflowed += st.overflowed; tot.exact_reruns += st.exact_reruns;
                tot.wide_launches = std::max(tot.wide_launches, st.wide_launches); tot.wide_queries = std::max(tot.wide_queries, st.wide_queries);
                tot.aux_cus = std::max(tot.aux_cus, st.aux_cus); tot.scans_overlap = std::max(tot.scans_overlap, st.scans_overlap);
                tot.scan_kernel = std::max(tot.scan_kernel, st.scan_kernel);
            }
        }
        if (r == VF_OK && !shard_is_staged(ix, g)) {
            hipError_t e = hipMemcpyPeerAsync((char*)gs.all.p + g * part, ix->device, gs.blob[g].p, sh->device, part, gs.feed[g]);
            if (e == hipSuccess) e = hipEventRecord(gs.ev[g], gs.feed[g]);
            if (e != hipSuccess) r = fail(VF_EHIP, std::string("sharded search: result peer copy: ") + hipGetErrorString(e));
            gs.staged_last[g] = 0;
        } else if (r == VF_OK) {   // shard -> pinned host on the shard's stream, host -> `all` on a home-device stream
            r = gs.hb[g].ensure(part);
            hipError_t e = hipSuccess;
            // hb[g] is free again once the PREVIOUS search's host -> home copy out of it has run
            if (r == VF_OK && gs.staged_last[g]) e = hipStreamWaitEvent(gs.feed[g], gs.ev_home[g], 0);
            if (r == VF_OK && e == hipSuccess) e = hipMemcpyAsync(gs.hb[g].p, gs.blob[g].p, part, hipMemcpyDeviceToHost, gs.feed[g]);
            if (r == VF_OK && e == hipSuccess) e = hipEventRecord(gs.ev_hb[g], gs.feed[g]);
            if (r == VF_OK && e == hipSuccess) e = hipSetDevice(ix->device);
            if (r == VF_OK && e == hipSuccess) e = hipStreamWaitEvent(gs.hfeed[g], gs.ev_hb[g], 0);
            if (r == VF_OK && e == hipSuccess) e = hipMemcpyAsync((char*)gs.all.p + g * part, gs.hb[g].p, part, hipMemcpyHostToDevice, gs.hfeed[g]);
            if (r == VF_OK && e == hipSuccess) e = hipEventRecord(gs.ev_home[g], gs.hfeed[g]);
            if (r == VF_OK && e != hipSuccess) r = fail(VF_EHIP, std::string("sharded search: staged result copy: ") + hipGetErrorString(e));
            if (r == VF_OK) gs.staged_last[g] = 1;
        }
        if (r != VF_OK && rc == VF_OK) rc = r;
    }
    if (rc != VF_OK) return rc;
    VF_HIP(hipSetDevice(ix->device));
    VF_HIP(scan_configure());
    for (size_t g = 0; g < G; ++g) VF_HIP(hipStreamWaitEvent(gs.user, shard_is_staged(ix, g) ? gs.ev_home[g] : gs.ev[g], 0));
    // parts are in ascending id-range order (shard order), which the merge's tie rule relies on
    VF_HIP(launch_merge_topk_packed(gs.all.p, (int)G, gs.nq, gs.k, (long long*)gs.d_ids, gs.d_scores, gs.user));
    ix->stats = tot;
    return VF_OK;
}

static int search_begin_locked(vf_index* ix, int slot, const float* d_queries, int nq, int k, int64_t* d_ids,
                               float* d_scores, hipStream_t stream) {
    if (!ix->shards.empty()) return group_begin(ix, slot, d_queries, nq, k, d_ids, d_scores, stream);
    VF_HIP(hipSetDevice(ix->device));
    int rc = begin_impl(ix, slot, d_queries, nq, k, d_ids, d_scores, stream);
    if (rc != VF_OK) ix->slots[slot].pending = false;
    return rc;
}

static int search_end_locked(vf_index* ix, int slot) {
    if (!ix->shards.empty()) return group_end(ix, slot);
    VF_HIP(hipSetDevice(ix->device));
    return end_impl(ix, slot);
}

static bool slot_pending(const vf_index* ix, int slot) {
    return ix->shards.empty() ? ix->slots[slot].pending : ix->gslots[slot].pending;
}

extern "C" int vf_index_search_begin(vf_index* ix, int32_t slot, const float* d_queries, int32_t nq, int32_t k,
                                     int64_t* d_ids, float* d_scores, void* stream) {
    DeviceGuard restore_callers_device;
    VF_TRY(check_search_args(ix, d_queries, nq, k, d_ids, d_scores));
    if (slot < 0 || slot >= kSlots) return fail(VF_EINVAL, "vf_index_search_begin: bad slot");
    std::lock_guard<std::mutex> g(ix->mu);
    return search_begin_locked(ix, slot, d_queries, nq, k, d_ids, d_scores, (hipStream_t)stream);
}

extern "C" int vf_index_search_end(vf_index* ix, int32_t slot) {
    DeviceGuard restore_callers_device;
    if (!ix) return fail(VF_EINVAL, "vf_index_search_end: null handle");
    if (slot < 0 || slot >= kSlots) return fail(VF_EINVAL, "vf_index_search_end: bad slot");
    std::lock_guard<std::mutex> g(ix->mu);
    return search_end_locked(ix, slot);
}

static int search_device_locked(vf_index* ix, const float* d_queries, int nq, int k, int64_t* d_ids, float* d_scores,
                                hipStream_t stream) {
    if (slot_pending(ix, 0)) return fail(VF_EINVAL, "vf_index_search_device: slot 0 busy (begin without end)");
    VF_TRY(search_begin_locked(ix, 0, d_queries, nq, k, d_ids, d_scores, stream));
    return search_end_locked(ix, 0);
}

extern "C" int vf_index_search_device(vf_index* ix, const float* d_queries, int32_t nq, int32_t k, int64_t* d_ids,
                                      float* d_scores, void* stream) {
    DeviceGuard restore_callers_device;
    VF_TRY(check_search_args(ix, d_queries, nq, k, d_ids, d_scores));
    std::lock_guard<std::mutex> g(ix->mu);
    return search_device_locked(ix, d_queries, nq, k, d_ids, d_scores, (hipStream_t)stream);
}

// Host buffers in and out: what FaissRetriever.invoke hands over (NumPy arrays, src/utils/faissRetriever.py:34-38).
// The device staging lives in the handle and is reused from call to call (no hipMalloc / hipFree per request); the
// handle's mutex is held for the whole call, so concurrent request threads take turns on it.
extern "C" int vf_index_search(vf_index* ix, const float* queries, int32_t nq, int32_t k, int64_t* out_ids,
                               float* out_scores) {
    DeviceGuard restore_callers_device;
    VF_TRY(check_search_args(ix, queries, nq, k, out_ids, out_scores));
    if (nq == 0 || k == 0) return VF_OK;
    std::lock_guard<std::mutex> g(ix->mu);
    VF_HIP(hipSetDevice(ix->device));
    const size_t qb = (size_t)nq * ix->d * sizeof(float), ib = (size_t)nq * k * sizeof(int64_t), sb = (size_t)nq * k * sizeof(float);
    VF_TRY(ix->st_q.ensure(qb));
    VF_TRY(ix->st_ids.ensure(ib));
    VF_TRY(ix->st_sc.ensure(sb));
    VF_HIP(hipMemcpy(ix->st_q.p, queries, qb, hipMemcpyHostToDevice));
    VF_TRY(search_device_locked(ix, ix->st_q.as<float>(), nq, k, ix->st_ids.as<int64_t>(), ix->st_sc.as<float>(), nullptr));
    VF_HIP(hipSetDevice(ix->device));
    VF_HIP(hipStreamSynchronize(nullptr));
    VF_HIP(hipMemcpy(out_ids, ix->st_ids.p, ib, hipMemcpyDeviceToHost));
    VF_HIP(hipMemcpy(out_scores, ix->st_sc.p, sb, hipMemcpyDeviceToHost));
    return VF_OK;
}

static std::atomic<bool> g_force_no_peer{false};
// Test hook (not in the public header): handles grouped from now on treat EVERY shard as unreachable by peer copy.
extern "C" int vf_debug_force_no_peer(int32_t on) {
    return g_force_no_peer.exchange(on != 0) ? 1 : 0;
}

// Take ownership of per-device indexes (built with id_offset = first row of their block, in ascending block order)
// and present them as one index.  home = shards[0]'s device.
static int make_group(vf_index** out, std::vector<vf_index*>& shards) {
    vf_index* grp = new (std::nothrow) vf_index();
    if (!grp) return fail(VF_ENOMEM, "vf_index_group: host allocation failed");
    grp->device = shards[0]->device;
    grp->d = shards[0]->d; grp->dp = shards[0]->dp; grp->dtype = shards[0]->dtype; grp->id_offset = shards[0]->id_offset;
    grp->n_cu = shards[0]->n_cu;
    grp->n = 0;
    for (vf_index* sh : shards) grp->n += sh->n;
    grp->shards = shards;
    // peer access home <-> shard devices.  A link that cannot be enabled is not fatal -- hipMemcpyPeerAsync then stages
    // through the host -- but it is RECORDED (vf_index_peer_access) so that the slow exchange does not go unnoticed.
    auto enable = [](int from, int to) -> bool {
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, from, to) != hipSuccess || !can) return false;
        if (hipSetDevice(from) != hipSuccess) return false;
        const hipError_t e = hipDeviceEnablePeerAccess(to, 0);
        (void)hipGetLastError();
        return e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled;
    };
    grp->peer_ok.assign(shards.size(), 1);
    const bool forced = g_force_no_peer.load(std::memory_order_relaxed);
    for (size_t g = 0; g < shards.size(); ++g) {
        // test hook: every shard -- the ones on the home device too -- takes the host-staged exchange, so that the path a refused
        // peer link falls to runs on a one-GPU box (device_ids = [0, 0]); nothing is enabled or disabled on the devices
        if (forced) { grp->peer_ok[g] = 0; continue; }
        if (shards[g]->device == grp->device) continue;
        const bool a = enable(grp->device, shards[g]->device), b = enable(shards[g]->device, grp->device);
        grp->peer_ok[g] = a && b ? 1 : 0;
    }
    (void)hipSetDevice(grp->device);
    *out = grp;
    return VF_OK;
}

extern "C" int vf_index_group(vf_index** out, vf_index** shards, int32_t n_shards) {
    DeviceGuard restore_callers_device;
    if (!out || !shards || n_shards <= 0) return fail(VF_EINVAL, "vf_index_group: bad argument");
    *out = nullptr;
    std::vector<vf_index*> v(shards, shards + n_shards);
    int64_t next = v[0] ? v[0]->id_offset : 0;
    for (vf_index* sh : v) {
        if (!sh || !sh->shards.empty()) return fail(VF_EINVAL, "vf_index_group: null or nested shard");
        if (sh->d != v[0]->d || sh->dtype != v[0]->dtype) return fail(VF_EINVAL, "vf_index_group: shards differ in d / dtype");
        if (sh->id_offset != next) return fail(VF_EINVAL, "vf_index_group: shards must be contiguous row blocks in ascending order");
        next += sh->n;
    }
    return make_group(out, v);
}

static int check_devices(const int32_t* device_ids, int32_t n_dev, const char* who) {
    if (!device_ids || n_dev <= 0 || n_dev > 64) return fail(VF_EINVAL, std::string(who) + ": bad device list");
    int ndev = 0;
    VF_HIP(hipGetDeviceCount(&ndev));
    for (int i = 0; i < n_dev; ++i)
        if (device_ids[i] < 0 || device_ids[i] >= ndev) return fail(VF_EINVAL, std::string(who) + ": bad device id");
    return VF_OK;
}

static void shard_block(int64_t n, int n_dev, int g, int64_t* lo, int64_t* hi) {  // SURVEY.md 8e partitioning
    const int64_t per = (n + n_dev - 1) / n_dev;
    *lo = std::min<int64_t>(n, (int64_t)g * per);
    *hi = std::min<int64_t>(n, *lo + per);
}

extern "C" int vf_index_create_sharded(vf_index** out, const void* rows, int64_t n, int32_t d, int32_t dtype,
                                       const int32_t* device_ids, int32_t n_dev) {
    DeviceGuard restore_callers_device;
    if (!out) return fail(VF_EINVAL, "vf_index_create_sharded: null out");
    *out = nullptr;
    VF_TRY(check_devices(device_ids, n_dev, "vf_index_create_sharded"));
    if (n < 0 || d <= 0 || (n > 0 && !rows)) return fail(VF_EINVAL, "vf_index_create_sharded: bad rows/n/d");
    if (dtype != VF_DTYPE_F32 && dtype != VF_DTYPE_F16 && dtype != VF_DTYPE_FP8_E4M3)
        return fail(VF_EINVAL, "vf_index_create_sharded: unknown dtype");
    const size_t esz = dtype == VF_DTYPE_F32 ? 4 : (dtype == VF_DTYPE_F16 ? 2 : 1);
    std::vector<vf_index*> shards;
    int rc = VF_OK;
    for (int g = 0; g < n_dev && rc == VF_OK; ++g) {
        int64_t lo, hi;
        shard_block(n, n_dev, g, &lo, &hi);
        vf_index* sh = nullptr;
        rc = create_impl(&sh, (const char*)rows + (size_t)lo * d * esz, false, hi - lo, d, dtype, device_ids[g], lo);
        if (rc == VF_OK) shards.push_back(sh);
    }
    if (rc == VF_OK) rc = make_group(out, shards);
    if (rc != VF_OK) { const std::string keep = g_err; for (vf_index* sh : shards) destroy_index(sh); g_err = keep; }
    return rc;
}

extern "C" int vf_index_create_sharded_from_file(vf_index** out, const char* path, const int32_t* device_ids, int32_t n_dev) {
    DeviceGuard restore_callers_device;
    if (!out || !path) return fail(VF_EINVAL, "vf_index_create_sharded_from_file: null argument");
    *out = nullptr;
    VF_TRY(check_devices(device_ids, n_dev, "vf_index_create_sharded_from_file"));
    int64_t n = 0;
    VF_TRY(vf_corpus_file_info(path, &n, nullptr, nullptr, nullptr));
    std::vector<vf_index*> shards;
    int rc = VF_OK;
    for (int g = 0; g < n_dev && rc == VF_OK; ++g) {
        int64_t lo, hi;
        shard_block(n, n_dev, g, &lo, &hi);
        vf_index* sh = nullptr;
        rc = vf_index_create_from_file(&sh, path, lo, hi, device_ids[g], lo);
        if (rc == VF_OK) shards.push_back(sh);
    }
    if (rc == VF_OK) rc = make_group(out, shards);
    if (rc != VF_OK) { const std::string keep = g_err; for (vf_index* sh : shards) destroy_index(sh); g_err = keep; }
    return rc;
}

extern "C" int vf_index_shards(vf_index* ix, int32_t* n_shards, int32_t* device_ids, int32_t cap) {
    if (!ix || !n_shards) return fail(VF_EINVAL, "vf_index_shards: null argument");
    const int G = (int)ix->shards.size();
    *n_shards = G;
    for (int g = 0; g < G && g < cap && device_ids; ++g) device_ids[g] = ix->shards[g]->device;
    return VF_OK;
}

extern "C" int vf_index_peer_access(vf_index* ix, int32_t* ok, int32_t cap, int32_t* n_missing) {
    if (!ix) return fail(VF_EINVAL, "vf_index_peer_access: null handle");
    int missing = 0;
    for (size_t g = 0; g < ix->peer_ok.size(); ++g) {
        if (ok && (int)g < cap) ok[g] = ix->peer_ok[g];
        missing += ix->peer_ok[g] ? 0 : 1;
    }
    if (n_missing) *n_missing = missing;
    return VF_OK;
}

// ------------------------------------------------------------------------------------------------
// small dense cosine (compute_similarity_mtx / cosine_similarity restatements)
//
// These entry points sit on the serve chain (rank_chunk calls two of them per request) and used to pay 3 - 8 hipMalloc / hipFree
// pairs per call.  They now lease ONE arena from a small process-wide pool: a call pops an arena of its device (or makes one),
// carves its buffers out of it and pushes it back -- concurrent request threads each hold their own, none waits for another, and
// nothing hangs on thread-exit destructors.  Arenas above 64 MB and arenas beyond eight go back to the driver on release.
// ------------------------------------------------------------------------------------------------
namespace {
struct Arena { int device = -1; DevBuf buf; };
std::mutex g_arena_mu;
std::vector<Arena*> g_arena_free;
std::atomic<long long> g_arena_allocs{0};            // hipMalloc calls made for arenas (test hook: vf_debug_small_allocs)
constexpr size_t kArenaKeepBytes = (size_t)64 << 20, kArenaKeepCount = 8;

struct Lease {
    Arena* a = nullptr;
    size_t used = 0;
    // the caller has made `device` current
    int acquire(int device, size_t bytes) {
        {
            std::lock_guard<std::mutex> lk(g_arena_mu);
            int best = -1;
            for (int i = 0; i < (int)g_arena_free.size(); ++i)
                if (g_arena_free[i]->device == device && (best < 0 || g_arena_free[i]->buf.bytes > g_arena_free[best]->buf.bytes)) best = i;
            if (best >= 0) { a = g_arena_free[best]; g_arena_free.erase(g_arena_free.begin() + best); }
        }
        if (!a) { a = new (std::nothrow) Arena(); if (!a) return fail(VF_ENOMEM, "small-op arena: host allocation failed"); a->device = device; }
        if (bytes > a->buf.bytes) g_arena_allocs.fetch_add(1, std::memory_order_relaxed);
        const int rc = a->buf.ensure(bytes);
        if (rc != VF_OK) { a->buf.release(); delete a; a = nullptr; }
        return rc;
    }
    static size_t padded(size_t bytes) { return (bytes + 255) & ~(size_t)255; }
    template <class T> T* take(size_t count) {
        T* p = (T*)((char*)a->buf.p + used);
        used += padded(count * sizeof(T));
        return p;
    }
    ~Lease() {
        if (!a) return;
        std::unique_lock<std::mutex> lk(g_arena_mu);
        if (a->buf.bytes <= kArenaKeepBytes && g_arena_free.size() < kArenaKeepCount) { g_arena_free.push_back(a); return; }
        lk.unlock();
        a->buf.release();
        delete a;
    }
};
}  // namespace

// Test hook (not in the public header): resident workgroups per CU of k_scan_wide8<waves> with `stage_cap` candidate-stage entries
extern "C" int vf_debug_wide8_occupancy(int32_t waves, int32_t stage_cap) {
    if (scan_configure() != hipSuccess) return -1;
    return scan_wide8_occupancy(waves, stage_cap > 0 ? stage_cap : scan_wide8_stage_cap(waves));
}

// Test hook (not in the public header): arena allocations so far -- steady-state calls of the small entry points make none.
extern "C" long long vf_debug_small_allocs(void) { return g_arena_allocs.load(std::memory_order_relaxed); }

extern "C" int vf_cosine_scores(const float* a, int32_t na, const float* b, int64_t nb, int32_t d, float* out,
                                int32_t device_id) {
    DeviceGuard restore_callers_device;
    if (na < 0 || nb < 0 || d <= 0) return fail(VF_EINVAL, "vf_cosine_scores: bad sizes");
    if (na == 0 || nb == 0) return VF_OK;
    if (!a || !b || !out) return fail(VF_EINVAL, "vf_cosine_scores: null buffer");
    VF_HIP(hipSetDevice(device_id));
    const bool same = a == b && (int64_t)na == nb;       // vf_cosine_matrix: one operand, prepared once
    const size_t ea = (size_t)na * d, eb = same ? 0 : (size_t)nb * d;
    Lease ws;
    VF_TRY(ws.acquire(device_id, 2 * Lease::padded(ea * 4) + 2 * Lease::padded(eb * 4) + Lease::padded((size_t)na * 4) + Lease::padded((size_t)nb * 4) +
                                     Lease::padded((size_t)std::max<int64_t>(na, nb) * 4) + Lease::padded((size_t)na * nb * 4)));
    float* da = ws.take<float>(ea); float* an = ws.take<float>(ea);
    float* db = same ? da : ws.take<float>(eb); float* bn = same ? an : ws.take<float>(eb);
    float* norm_a = ws.take<float>(na); float* norm_b = same ? norm_a : ws.take<float>(nb);
    float* tmp = ws.take<float>(std::max<int64_t>(na, nb)); float* dout = ws.take<float>((size_t)na * nb);
    VF_HIP(hipMemcpy(da, a, ea * 4, hipMemcpyHostToDevice));
    if (!same) VF_HIP(hipMemcpy(db, b, eb * 4, hipMemcpyHostToDevice));
    VF_HIP(launch_prep_rows(da, 0, na, d, d, nullptr, norm_a, tmp, nullptr));
    if (!same) VF_HIP(launch_prep_rows(db, 0, nb, d, d, nullptr, norm_b, tmp, nullptr));
    VF_HIP(launch_normalize_rows(da, 0, 0, na, d, norm_a, an, nullptr));
    if (!same) VF_HIP(launch_normalize_rows(db, 0, 0, nb, d, norm_b, bn, nullptr));
    VF_HIP(launch_dense_dot16(an, na, bn, nb, d, dout, nb, nullptr));
    VF_HIP(hipMemcpy(out, dout, (size_t)na * nb * 4, hipMemcpyDeviceToHost));
    return VF_OK;
}

extern "C" int vf_cosine_matrix(const float* x, int32_t n, int32_t d, float* out, int32_t device_id) {
    return vf_cosine_scores(x, n, x, n, d, out, device_id);
}

// rows `sel` (local row numbers) of ONE device's index, canonically normalised, into xn [count][d] on that device (null stream)
static int gather_normalised(vf_index* sh, Lease& ws, const std::vector<long long>& sel, float* xn) {
    long long* dsel = ws.take<long long>(sel.size());
    VF_HIP(hipMemcpy(dsel, sel.data(), sel.size() * 8, hipMemcpyHostToDevice));
    VF_HIP(launch_normalize_rows_gather(sh->rows_orig, sh->dtype, dsel, (int)sel.size(), sh->d, sh->norm, xn, nullptr));
    return VF_OK;
}

// Cosine matrix of rows ALREADY in the index, picked by id (round 4): compute_similarity_mtx re-embeds the n retrieved chunk
// texts (src/utils/ensembleRetriever.py:265-281: n encoder forwards upstream, one batched forward here -- 19 ms for 100 chunks
// of 512 tokens); when the chunks came out of this index and the embedder treats documents and queries alike, their
// embeddings are rows of the HBM-resident corpus: gather + canonical normalise (the index's own norms) + dot16.
// A SHARDED handle (round 5): every shard normalises its own rows on its own device (the norms are per row, so the values are the
// single-device ones bit for bit), the blocks travel to the home device -- peer copy, or through the host where the link is
// missing -- and are put back into the caller's order there by the same gather kernel (norm 1: x * (float)(1.0 / 1.0) = x).
// Mixed form (round 6): ids[i] >= 0 picks a corpus row as above; ids[i] == -1 takes the next vector of `extra` ([n_extra][d] fp32
// host, in order of appearance) -- a chunk text the corpus does not hold, embedded by the caller.  Those vectors get the canonical
// norm of vf_cosine_matrix (k_prep_rows on fp32 values), so an all-extra call returns vf_cosine_matrix's bits and an all-row call
// vf_cosine_matrix_rows's.  This is what lets EnsembleRetriever.compute_similarity_mtx(texts) -- the reference's call, texts only
// (src/utils/vllmManager.py:462) -- serve known texts from HBM and embed only the unknown ones.
static int cosine_matrix_rows_impl(vf_index* ix, const int64_t* ids, int32_t n, const float* extra, int32_t n_extra, float* out,
                                   const char* who) {
    DeviceGuard restore_callers_device;
    if (!ix) return fail(VF_EINVAL, std::string(who) + ": null handle");
    if (n < 0 || n_extra < 0) return fail(VF_EINVAL, std::string(who) + ": negative n");
    if (n == 0) return VF_OK;
    if (!ids || !out) return fail(VF_EINVAL, std::string(who) + ": null buffer");
    if (n > 4096) return fail(VF_EINVAL, std::string(who) + ": n must be <= 4096");
    if (n_extra > 0 && !extra) return fail(VF_EINVAL, std::string(who) + ": n_extra > 0 with a null extra buffer");
    const size_t nd = (size_t)n * ix->d;
    std::lock_guard<std::mutex> lk(ix->mu);
    int n_minus = 0;
    for (int i = 0; i < n; ++i) if (ids[i] == -1) ++n_minus;
    if (n_minus != n_extra) return fail(VF_EINVAL, std::string(who) + ": the ids hold " + std::to_string(n_minus) + " entries of -1 but n_extra is " + std::to_string(n_extra));
    if (ix->shards.empty() && n_extra == 0) {
        std::vector<long long> sel((size_t)n);
        for (int i = 0; i < n; ++i) {
            const int64_t r = ids[i] - ix->id_offset;
            if (r < 0 || r >= ix->n) return fail(VF_EINVAL, std::string(who) + ": id outside the index");
            sel[(size_t)i] = r;
        }
        VF_HIP(hipSetDevice(ix->device));
        Lease ws;
        VF_TRY(ws.acquire(ix->device, Lease::padded((size_t)n * 8) + Lease::padded(nd * 4) + Lease::padded((size_t)n * n * 4)));
        float* xn = ws.take<float>(nd); float* dout = ws.take<float>((size_t)n * n);
        VF_TRY(gather_normalised(ix, ws, sel, xn));
        VF_HIP(launch_dense_dot16(xn, n, xn, n, ix->d, dout, n, nullptr));
        VF_HIP(hipMemcpy(out, dout, (size_t)n * n * 4, hipMemcpyDeviceToHost));
        return VF_OK;
    }
    // general form: the blocks of the G shards (a plain handle is its own only shard), then the block of the extra vectors, meet in
    // `cat` on the home device and are put into the caller's order by one gather
    std::vector<vf_index*> parts = ix->shards;
    if (parts.empty()) parts.push_back(ix);
    const size_t G = parts.size();
    std::vector<std::vector<long long>> sel(G);
    std::vector<int> owner((size_t)n), rank_in((size_t)n);
    int seen_extra = 0;
    for (int i = 0; i < n; ++i) {
        if (ids[i] == -1) { owner[(size_t)i] = (int)G; rank_in[(size_t)i] = seen_extra++; continue; }
        size_t g = 0;
        while (g < G && !(ids[i] >= parts[g]->id_offset && ids[i] < parts[g]->id_offset + parts[g]->n)) ++g;
        if (g == G) return fail(VF_EINVAL, std::string(who) + ": id outside the index");
        owner[(size_t)i] = (int)g; rank_in[(size_t)i] = (int)sel[g].size();
        sel[g].push_back(ids[i] - parts[g]->id_offset);
    }
    std::vector<long long> start(G + 2, 0), where((size_t)n);   // block of shard g in the concatenation (block G: the extras); position i's row in it
    for (size_t g = 0; g < G; ++g) start[g + 1] = start[g] + (long long)sel[g].size();
    start[G + 1] = start[G] + n_extra;
    for (int i = 0; i < n; ++i) where[(size_t)i] = start[(size_t)owner[(size_t)i]] + rank_in[(size_t)i];
    const size_t ed = (size_t)n_extra * ix->d;
    VF_HIP(hipSetDevice(ix->device));
    Lease home;
    VF_TRY(home.acquire(ix->device, 2 * Lease::padded(nd * 4) + Lease::padded((size_t)n * 4) + Lease::padded((size_t)n * 8) + Lease::padded((size_t)n * n * 4) +
                                    Lease::padded(ed * 4) + 2 * Lease::padded((size_t)n_extra * 4)));
    float* cat = home.take<float>(nd); float* xn = home.take<float>(nd); float* ones = home.take<float>(n);
    long long* dwhere = home.take<long long>(n); float* dout = home.take<float>((size_t)n * n);
    std::vector<float> stage;
    for (size_t g = 0; g < G; ++g) {
        if (sel[g].empty()) continue;
        vf_index* sh = parts[g];
        const size_t cnt = sel[g].size(), bytes = cnt * ix->d * 4;
        float* dst = cat + (size_t)start[g] * ix->d;
        VF_HIP(hipSetDevice(sh->device));
        if (sh->device == ix->device && !shard_is_staged(ix, g)) {        // already at home: normalise straight into its block
            Lease ws;
            VF_TRY(ws.acquire(sh->device, Lease::padded(cnt * 8)));
            VF_TRY(gather_normalised(sh, ws, sel[g], dst));
            VF_HIP(hipStreamSynchronize(nullptr));
            continue;
        }
        Lease ws;
        VF_TRY(ws.acquire(sh->device, Lease::padded(cnt * 8) + Lease::padded(bytes)));
        float* part = ws.take<float>(cnt * ix->d);
        VF_TRY(gather_normalised(sh, ws, sel[g], part));
        VF_HIP(hipStreamSynchronize(nullptr));
        if (!shard_is_staged(ix, g)) {
            VF_HIP(hipMemcpyPeer(dst, ix->device, part, sh->device, bytes));
        } else {                                                          // no peer link: through the host
            stage.resize(cnt * ix->d);
            VF_HIP(hipMemcpy(stage.data(), part, bytes, hipMemcpyDeviceToHost));
            VF_HIP(hipSetDevice(ix->device));
            VF_HIP(hipMemcpy(dst, stage.data(), bytes, hipMemcpyHostToDevice));
        }
    }
    VF_HIP(hipSetDevice(ix->device));
    if (n_extra > 0) {   // the caller's vectors: canonical norms of their fp32 values (vf_cosine_matrix's arithmetic), normalised into block G
        float* raw = home.take<float>(ed); float* norm_e = home.take<float>(n_extra); float* tmp = home.take<float>(n_extra);
        VF_HIP(hipMemcpy(raw, extra, ed * 4, hipMemcpyHostToDevice));
        VF_HIP(launch_prep_rows(raw, 0, n_extra, ix->d, ix->d, nullptr, norm_e, tmp, nullptr));
        VF_HIP(launch_normalize_rows(raw, 0, 0, n_extra, ix->d, norm_e, cat + (size_t)start[G] * ix->d, nullptr));
    }
    const std::vector<float> one((size_t)n, 1.0f);
    VF_HIP(hipMemcpy(ones, one.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    VF_HIP(hipMemcpy(dwhere, where.data(), (size_t)n * 8, hipMemcpyHostToDevice));
    VF_HIP(launch_normalize_rows_gather(cat, VF_DTYPE_F32, dwhere, n, ix->d, ones, xn, nullptr));
    VF_HIP(launch_dense_dot16(xn, n, xn, n, ix->d, dout, n, nullptr));
    VF_HIP(hipMemcpy(out, dout, (size_t)n * n * 4, hipMemcpyDeviceToHost));
    return VF_OK;
}

extern "C" int vf_cosine_matrix_rows(vf_index* ix, const int64_t* ids, int32_t n, float* out) {
    if (ids && n > 0 && n <= 4096)   // -1 is the mixed form's marker; here every id must name a row
        for (int i = 0; i < n; ++i) if (ids[i] == -1) return fail(VF_EINVAL, "vf_cosine_matrix_rows: id outside the index");
    return cosine_matrix_rows_impl(ix, ids, n, nullptr, 0, out, "vf_cosine_matrix_rows");
}

extern "C" int vf_cosine_matrix_rows_mixed(vf_index* ix, const int64_t* ids, int32_t n, const float* extra, int32_t n_extra, float* out) {
    return cosine_matrix_rows_impl(ix, ids, n, extra, n_extra, out, "vf_cosine_matrix_rows_mixed");
}

extern "C" int vf_merge_topk_device(const int64_t* d_ids_parts, const float* d_score_parts, int32_t nparts, int32_t nq,
                                    int32_t k, int64_t* d_ids, float* d_scores, int32_t device_id, void* stream) {
    DeviceGuard restore_callers_device;
    if (nparts <= 0 || nq < 0 || k < 0) return fail(VF_EINVAL, "vf_merge_topk_device: bad sizes");
    if (nq == 0 || k == 0) return VF_OK;
    if (!d_ids_parts || !d_score_parts || !d_ids || !d_scores) return fail(VF_EINVAL, "vf_merge_topk_device: null buffer");
    if ((size_t)nparts * k > 16384) return fail(VF_EUNSUPPORTED, "vf_merge_topk_device: nparts * k > 16384");
    VF_HIP(hipSetDevice(device_id));
    VF_HIP(scan_configure());
    VF_HIP(launch_merge_topk((const long long*)d_ids_parts, d_score_parts, nparts, nq, k, (long long*)d_ids, d_scores,
                             (hipStream_t)stream));
    return VF_OK;
}

extern "C" int vf_merge_topk_packed_device(const void* d_parts, int32_t nparts, int32_t nq, int32_t k, int64_t* d_ids,
                                           float* d_scores, int32_t device_id, void* stream) {
    DeviceGuard restore_callers_device;
    if (nparts <= 0 || nq < 0 || k < 0) return fail(VF_EINVAL, "vf_merge_topk_packed_device: bad sizes");
    if (nq == 0 || k == 0) return VF_OK;
    if (!d_parts || !d_ids || !d_scores) return fail(VF_EINVAL, "vf_merge_topk_packed_device: null buffer");
    if ((size_t)nparts * k > 16384) return fail(VF_EUNSUPPORTED, "vf_merge_topk_packed_device: nparts * k > 16384");
    VF_HIP(hipSetDevice(device_id));
    VF_HIP(scan_configure());
    VF_HIP(launch_merge_topk_packed(d_parts, nparts, nq, k, (long long*)d_ids, d_scores, (hipStream_t)stream));
    return VF_OK;
}

extern "C" int vf_fuse_rank(const float* rerank_scores, const float* time_scores, int32_t n, float* out_scores,
                            int64_t* out_order, int32_t device_id) {
    DeviceGuard restore_callers_device;
    if (n < 0 || n > 4096) return fail(VF_EINVAL, "vf_fuse_rank: n must be in [0, 4096]");
    if (n == 0) return VF_OK;
    if (!rerank_scores || !time_scores || !out_scores || !out_order) return fail(VF_EINVAL, "vf_fuse_rank: null buffer");
    VF_HIP(hipSetDevice(device_id));
    Lease ws;
    VF_TRY(ws.acquire(device_id, 3 * Lease::padded((size_t)n * 4) + Lease::padded((size_t)n * 8)));
    float* a = ws.take<float>(n); float* b = ws.take<float>(n); float* o = ws.take<float>(n); long long* ord = ws.take<long long>(n);
    VF_HIP(hipMemcpy(a, rerank_scores, (size_t)n * 4, hipMemcpyHostToDevice));
    VF_HIP(hipMemcpy(b, time_scores, (size_t)n * 4, hipMemcpyHostToDevice));
    VF_HIP(launch_fuse_rank(a, b, n, o, ord, nullptr));
    VF_HIP(hipMemcpy(out_scores, o, (size_t)n * 4, hipMemcpyDeviceToHost));
    VF_HIP(hipMemcpy(out_order, ord, (size_t)n * 8, hipMemcpyDeviceToHost));
    return VF_OK;
}

// Test hook (not in the public header): the fused scan's HARDWARE e4m3 -> fp16 conversion applied to `count` codes,
// so that tests can pin it against the oracle's table instead of trusting the instruction's documentation.
extern "C" int vf_debug_cvt_e4m3(const unsigned char* codes, float* out, int32_t count) {
    DeviceGuard restore_callers_device;
    if (!codes || !out || count < 0) return fail(VF_EINVAL, "vf_debug_cvt_e4m3: bad argument");
    unsigned char* d_in = nullptr; float* d_out = nullptr;
    VF_HIP(hipMalloc((void**)&d_in, (size_t)count + 8));
    hipError_t e = hipMalloc((void**)&d_out, ((size_t)count + 8) * 4);
    if (e == hipSuccess) e = hipMemcpy(d_in, codes, count, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = launch_debug_cvt_e4m3(d_in, d_out, count, nullptr);
    if (e == hipSuccess) e = hipMemcpy(out, d_out, (size_t)count * 4, hipMemcpyDeviceToHost);
    (void)hipFree(d_in); (void)hipFree(d_out);
    if (e != hipSuccess) return fail(VF_EHIP, std::string("vf_debug_cvt_e4m3: ") + hipGetErrorString(e));
    return VF_OK;
}
