// vf_internal.h -- shared between the kernel TU (vf_kernels.hip) and the C-ABI TU (vf_api.hip).
// gfx950 only.  Not part of the public ABI (that is include/veritasfi_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/veritasfi_hip.h"  // VF_DTYPE_*

#include <string>

namespace vf {

// records the thread-local message vf_last_error() returns; returns `code` (defined in vf_api.hip)
int set_error(int code, const std::string& msg);

typedef unsigned long long u64;
typedef unsigned int u32;

// Every entry point that selects a device (hipSetDevice is per-thread state) puts the caller's device back on return:
// a host application that drives its own HIP / torch work on another device must not find it changed by a vf_* call.
struct DeviceGuard {
    int prev = -1;
    DeviceGuard() { if (hipGetDevice(&prev) != hipSuccess) prev = -1; }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

// ---- fixed design constants ------------------------------------------------------
constexpr int kQueryTile = 32;        // queries per MFMA N-tile (v_mfma_f32_32x32x16_f16)
constexpr int kMaxBatch = 64;         // queries per scan pass (2 N-tiles); larger nq loops
constexpr int kRowTile = 32;          // corpus rows per MFMA M-tile
constexpr int kScanThreads = 512;     // 8 waves per workgroup, one workgroup per CU
constexpr int kHistBins = 2048;       // threshold histogram over cosine in [-1, 1]
constexpr int kSmallN = 16384;        // <= this many rows: exact dense path (LDS sort)
constexpr int kMaxCap = 8192;         // candidate slots per query in the fused path (u64 each)
constexpr int kCntStride = 32;        // u32 elements between per-query counters: one 128-B line each
constexpr int kMaxKFused = 2048;      // largest k the fused path serves (k' <= 4096 <= cap/2)

enum ScanMode { kModeSample = 0, kModeMain = 1 };

// Arguments of the fused scan kernel (k_scan).  Plain struct passed by value.
struct ScanArgs {
    const char* rows;        // fp16 scan copy, row-major, row_bytes per row (dp * 2)
    const float* inv_scan;   // [n] 1 / (canonical norm * row scale): approx score = acc * inv_scan
    const _Float16* qimg;    // query image [dp/8][QN][8] fp16 (normalised queries, zero padded)
    long long n;             // rows in this shard
    int dp;                  // padded dim, multiple of 64
    long long row_bytes;
    int total_waves;         // TW: rows are split evenly over TW waves
    int samp;                // sample rows at the head of each wave's range
    // sample mode output
    float* s0;               // [QN][TW * samp] approx scores of the sample rows (-inf = empty)
    const long long* wg_base; // [grid] first row of each scan workgroup's range (n * wg / grid)
    // main mode state (all zeroed / seeded per batch)
    u32* cnt;                // [QN * kCntStride] candidates appended (one counter per 128-B line)
    int* tau_bin;            // [QN] current threshold bin (monotone non-decreasing)
    u32* hist;               // [QN][kHistBins] counts of appended candidates per bin
    u32* hist_coarse;        // [QN][64] the same counts per 32 fine bins
    int stage_cap;           // LDS candidate-stage entries per workgroup (main mode)
    u64* cand;               // [QN][cap] (orderkey(approx) << 32) | local row
    int cap;
    int kprime;              // k + margin: the threshold keeps >= kprime rows above it
    int refresh_every;       // recompute tau when a query's count crosses a multiple of this
    int nq;                  // real queries (<= QN); padded queries never pass
    u32* tile_cnt;           // [grid] pool counters: tiles claimed from the shared tail of each workgroup's row range (k_sel0 zeroes them); null = no stealing
    int scan_grid;           // workgroups of the main scan (k_sel0 resets that many counters)
    // wide scan (k_scan_wide): queries in the global image, 256-query tiles per pass, row groups
    int qn_total, jtiles, rgroups;
    u32* sib;                // [rgroups][4] super-tiles finished by each query-tile workgroup of a row group (zeroed per launch), or null
    int sib_slack;           // a workgroup starts super-tile t + 1 once every sibling has finished t - sib_slack
    unsigned long long* dbg; // optional [grid][8 waves][4] wall-clock stamps (debug bit 7), else null
    int debug;               // bit 0: timing experiment -- seed tau so that nothing passes (results invalid)
};

struct FinalArgs {
    const u32* cnt; const u64* cand; int cap;
    int top_cap, sel_cap;    // LDS areas of k_final (set by launch_final)
    const int* tau_bin;      // final thresholds of the scan (validity check)
    const void* rows_orig; int orig_dtype; long long orig_row_elems;  // exact rows for rescoring (VF_DTYPE_*)
    const float* norm;       // canonical norms [n]
    const float* qn;         // canonical normalised queries [nq][d] fp32
    int d; int k; int kprime; float eps;
    const float* eps_q;      // optional [nq] per-query certificate bound (k_scan_wide8: the query's own quantisation residual); null = eps
    long long n_rows;        // rows in the shard (certificate is moot when all were re-scored)
    long long id_offset;
    long long* out_ids; float* out_scores;   // [nq][k]
    int* flags;              // [nq] 0 = certified exact, 1 = uncertified, 2 = overflow
    u32* cand_count_out;     // [nq] copy of cnt for stats
    unsigned long long* dbg; // optional [nq][8] wall-clock stamps of the phases (debug), else null
};

// ---- launchers (defined in vf_kernels.hip) -----------------------------------------------------
hipError_t launch_prep_rows(const void* rows, int dt /* VF_DTYPE_* */, long long n, int d, int dp,
                            void* scan /*fp16 [n][dp] (fp8 rows: bytes [n][dp]); null when rows are used in place*/,
                            float* norm, float* inv_scan, hipStream_t s);
hipError_t launch_prep_queries(const float* q, int nq, int d, int dp, int qn_tile /*32 or 64*/,
                               float* qn, _Float16* qimg, hipStream_t s);
hipError_t launch_normalize_rows(const void* rows, int dt /* VF_DTYPE_* */, long long row0, long long nrows, int d,
                                 const float* norm, float* out, hipStream_t s);
hipError_t launch_normalize_rows_gather(const void* rows, int dt, const long long* sel /*device, local row numbers*/, int nsel, int d,
                                        const float* norm, float* out, hipStream_t s);
hipError_t launch_dense_dot16(const float* qn, int nq, const float* cn, long long nrows, int d,
                              float* out, long long out_stride, hipStream_t s);
hipError_t launch_sort_rows(const float* scores, long long score_stride, int nq, int n, int k,
                            long long id_base, long long* out_ids, float* out_scores, int out_stride,
                            hipStream_t s);
hipError_t launch_scan(const ScanArgs& a, int mode, int qn_tile, int grid, int want_g /*0 = auto*/,
                       int rows_are_fp8, hipStream_t s);
// test hook: the scan's hardware e4m3 -> fp16 conversion over `count` codes (device pointers)
hipError_t launch_debug_cvt_e4m3(const unsigned char* in, float* out, int count, hipStream_t s);
// k_scan2: the main scan with whole-line LDS-DMA corpus loads (fp16 or e4m3 rows); a.stage_cap = scan2_stage_cap(...) >= 256
hipError_t launch_scan2(const ScanArgs& a, int qn_tile, int grid, int rows_are_fp8, hipStream_t s);
size_t scan2_lds_bytes(int dp, int qn_tile, int stage_cap);
int scan2_stage_cap(int dp, int qn_tile, int rows_are_fp8);
// k_scan2r: k_scan2 with part of the query image in registers and deeper rings (fp16 rows of 768 elements, e4m3 rows of 768 / 1024); scan2r_stage_cap < 256 = not this kernel
size_t scan2r_lds_bytes(int dp, int qn_tile, int stage_cap, int f8);
int scan2r_stage_cap(int dp, int qn_tile, int f8);
hipError_t launch_scan2r(const ScanArgs& a, int qn_tile, int grid, int f8, hipStream_t s);
hipError_t launch_scan2r_sample(const ScanArgs& a, int qn_tile, int grid, int f8, hipStream_t s);
hipError_t launch_scan_wide(const ScanArgs& a, int mode, int rows_are_fp8, hipStream_t s);
size_t scan_wide_lds_bytes(int stage_cap);
// k_scan_wide8: the wide main scan on the fp8 matrix instruction (e4m3 rows; a.qimg = the hi / lo code image of launch_prep_wide8)
hipError_t launch_prep_wide8(const float* qn, int nq, int d, int dp, int qtot, unsigned char* img8, float* eps_q, hipStream_t s);
hipError_t launch_scan_wide8(const ScanArgs& a, int waves /* 8: 256-query tiles, one workgroup per CU; 4: 128-query tiles, two per CU */, hipStream_t s);
int scan_wide8_stage_cap(int waves);
int scan_wide8_occupancy(int waves, int stage_cap);
// k_scan2<NT, 2>: the narrow main scan on the fp8 matrix instruction (launch_scan2 with rows_are_fp8 = 2; a.qimg = this image)
hipError_t launch_sel0(const ScanArgs& a, int qn_tile, hipStream_t s);
hipError_t launch_final(FinalArgs a, int nq, hipStream_t s);
hipError_t launch_merge_topk(const long long* ids_parts, const float* score_parts, int nparts, int nq,
                             int k, long long* ids, float* scores, hipStream_t s);
hipError_t launch_merge_topk_packed(const void* parts, int nparts, int nq, int k, long long* ids, float* scores,
                                    hipStream_t s);
hipError_t launch_fuse_rank(const float* a, const float* b, int n, float* out, long long* order,
                            hipStream_t s);
// bytes of one packed per-shard result [ids nq*k int64][scores nq*k fp32], padded to 16 so that every part of an
// all-gathered buffer keeps its int64 ids 8-byte aligned (nq*k odd would otherwise put part 1 on a 4-byte boundary)
inline long long packed_part_bytes(int nq, int k) { return ((long long)nq * k * 12 + 15) / 16 * 16; }
size_t scan_lds_bytes(int dp, int qn_tile);
int scan_stage_cap(int dp, int qn_tile);
hipError_t scan_configure();   // sets max dynamic LDS on the scan kernels (once per process/device)

}  // namespace vf
