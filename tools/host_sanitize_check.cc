// Host-side sanitizer check of the C ABI (round 5; CPU box, no GPU needed): this driver is linked against vf_api.hip's HOST pass
// compiled with -fsanitize=address,undefined (tests/test_host_sanitized.py builds and runs it) and walks the entry points whose
// work happens BEFORE the first HIP call -- the .vfc header / payload parser (vf_corpus_file_info, vf_index_create_from_file,
// vf_index_create_sharded_from_file), the argument checks of the index / small-dense / sharded-handle entry points and the weight-size
// calculators -- with hostile inputs: truncated and oversized files, n x d products that wrap 64 bits, bad dtypes and versions,
// id-table flags without a table, row ranges outside the file, null pointers, absurd configurations, 4000 random mutations of valid
// headers.  Every call must come back with a code (never crash, never read out of bounds); VF_OK answers must be consistent with the
// file on disk.  On a box without a GPU the device-side paths end in VF_EINVAL / VF_EHIP ("bad device_id"), which is the point: the
// host logic in front of them is what is being exercised.  GPU sanitizers are not available on this pool (and not attempted).
#include "../include/veritasfi_hip.h"

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>

#include <random>
#include <string>
#include <vector>

static int g_fail = 0;
#define CHECK(cond, ...) do { if (!(cond)) { fprintf(stderr, "CHECK FAILED %s:%d: ", __FILE__, __LINE__); fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); ++g_fail; } } while (0)

struct Header { char magic[8]; uint32_t version, dtype; uint64_t n; uint32_t d, flags; uint8_t reserved[32]; };
static_assert(sizeof(Header) == 64, "header");

static std::string g_dir;
static std::string write_file(const char* name, const Header& h, size_t payload_bytes, size_t header_bytes = 64) {
    const std::string p = g_dir + "/" + name;
    FILE* f = fopen(p.c_str(), "wb");
    if (!f) { perror("fopen"); exit(2); }
    fwrite(&h, 1, header_bytes, f);
    std::vector<unsigned char> z(payload_bytes, 0x3c);
    if (payload_bytes) fwrite(z.data(), 1, payload_bytes, f);
    fclose(f);
    return p;
}
static Header good(uint32_t dtype, uint64_t n, uint32_t d, uint32_t flags = 0) {
    Header h{};
    memcpy(h.magic, "VFCORPUS", 8);
    h.version = 1; h.dtype = dtype; h.n = n; h.d = d; h.flags = flags;
    return h;
}
static size_t esz(uint32_t dt) { return dt == 0 ? 4 : dt == 1 ? 2 : 1; }

// one probe of a file through every entry point that parses it; returns vf_corpus_file_info's code
static int probe(const std::string& path, long long file_bytes) {
    int64_t n = -7; int32_t d = -7, dt = -7, has = -7;
    const int rc = vf_corpus_file_info(path.c_str(), &n, &d, &dt, &has);
    if (rc == VF_OK) {
        CHECK(n >= 0 && d > 0 && dt >= 0 && dt <= 2 && (has == 0 || has == 1), "accepted header with n %lld d %d dtype %d", (long long)n, d, dt);
        const long double need = 64.0L + (long double)n * d * esz((uint32_t)dt) + (has ? (long double)n * 8 : 0);
        CHECK(need <= (long double)file_bytes, "accepted a header needing %.0Lf bytes in a file of %lld", need, file_bytes);
    } else {
        CHECK(rc == VF_EINVAL || rc == VF_EUNSUPPORTED, "unexpected code %d", rc);
        CHECK(strlen(vf_last_error()) > 0, "no error message");
    }
    CHECK(vf_corpus_file_info(path.c_str(), nullptr, nullptr, nullptr, nullptr) == rc, "null outputs changed the answer");
    vf_index* ix = (vf_index*)0x1;
    const int64_t ranges[][2] = {{0, n > 0 ? n : 0}, {-1, 1}, {2, 1}, {0, (n > 0 ? n : 0) + 1}, {0, 0}, {INT64_MAX - 1, INT64_MAX}};
    for (auto& r : ranges) {
        ix = (vf_index*)0x1;
        const int c = vf_index_create_from_file(&ix, path.c_str(), r[0], r[1], 0, 0);
        if (c == VF_OK) { CHECK(ix != nullptr && ix != (vf_index*)0x1, "OK without a handle"); vf_index_destroy(ix); }   // (a GPU box)
        else CHECK(ix == nullptr, "failed call left a handle");
        if (rc != VF_OK) CHECK(c != VF_OK, "the loader accepted a file the parser refuses");
    }
    const int32_t devs[2] = {0, 0};
    ix = (vf_index*)0x1;
    const int c2 = vf_index_create_sharded_from_file(&ix, path.c_str(), devs, 2);
    if (c2 == VF_OK) vf_index_destroy(ix); else CHECK(ix == nullptr, "failed sharded call left a handle");
    return rc;
}

int main(int argc, char** argv) {
    char tmpl[] = "/tmp/vf_san_XXXXXX";
    g_dir = argc > 1 ? argv[1] : mkdtemp(tmpl);
    CHECK(vf_version() > 0, "version");
    // ---- files that must be refused
    CHECK(vf_corpus_file_info(nullptr, nullptr, nullptr, nullptr, nullptr) == VF_EINVAL, "null path");
    CHECK(vf_corpus_file_info((g_dir + "/nope.vfc").c_str(), nullptr, nullptr, nullptr, nullptr) == VF_EINVAL, "missing file");
    CHECK(probe(write_file("empty.vfc", good(0, 0, 4), 0, 0), 0) == VF_EINVAL, "empty file");
    CHECK(probe(write_file("short.vfc", good(0, 1, 4), 0, 63), 63) == VF_EINVAL, "63-byte file");
    { Header h = good(0, 1, 4); h.magic[0] = 'X'; CHECK(probe(write_file("magic.vfc", h, 16), 80) == VF_EINVAL, "bad magic"); }
    { Header h = good(0, 1, 4); h.version = 2; CHECK(probe(write_file("ver.vfc", h, 16), 80) == VF_EUNSUPPORTED, "version 2"); }
    { Header h = good(3, 1, 4); CHECK(probe(write_file("dtype.vfc", h, 16), 80) == VF_EINVAL, "dtype 3"); }
    { Header h = good(0xffffffffu, 1, 4); CHECK(probe(write_file("dtype2.vfc", h, 16), 80) == VF_EINVAL, "dtype 2^32-1"); }
    CHECK(probe(write_file("d0.vfc", good(0, 5, 0), 64), 128) == VF_EINVAL, "d = 0");
    CHECK(probe(write_file("trunc.vfc", good(0, 100, 8), 100 * 8 * 4 - 1), 64 + 3199) == VF_EINVAL, "payload one byte short");
    CHECK(probe(write_file("noids.vfc", good(1, 10, 8, 1), 10 * 8 * 2), 64 + 160) == VF_EINVAL, "id flag without the table");
    // n x d x element size wraps 64 bits: 2^61 x 16 x 4 = 2^67 -> 0 (+ 64): used to PASS the size check
    CHECK(probe(write_file("wrap1.vfc", good(0, 1ull << 61, 16), 4096), 64 + 4096) == VF_EINVAL, "n x d wraps to zero");
    CHECK(probe(write_file("wrap2.vfc", good(2, 0xffffffffffffffffull, 0xffffffffu), 4096), 64 + 4096) == VF_EINVAL, "n = 2^64-1");
    CHECK(probe(write_file("wrap3.vfc", good(1, (1ull << 63) + 3, 1, 1), 4096), 64 + 4096) == VF_EINVAL, "id table wraps");
    CHECK(probe(write_file("wrap4.vfc", good(0, (1ull << 62) / 3, 12), 4096), 64 + 4096) == VF_EINVAL, "n x d x 4 just past 2^64");
    // ---- files that must be accepted (by the parser; the loader then needs a GPU)
    for (uint32_t dt = 0; dt <= 2; ++dt)
        for (uint32_t fl = 0; fl <= 1; ++fl) {
            const uint64_t n = 37; const uint32_t d = 24;
            const size_t pay = n * d * esz(dt) + (fl ? n * 8 : 0);
            char name[64]; snprintf(name, sizeof name, "ok_%u_%u.vfc", dt, fl);
            CHECK(probe(write_file(name, good(dt, n, d, fl), pay), 64 + (long long)pay) == VF_OK, "valid file dtype %u flags %u", dt, fl);
            snprintf(name, sizeof name, "ok_long_%u_%u.vfc", dt, fl);
            CHECK(probe(write_file(name, good(dt, n, d, fl), pay + 1000), 64 + (long long)pay + 1000) == VF_OK, "valid file with trailing bytes");
        }
    CHECK(probe(write_file("n0.vfc", good(0, 0, 8), 0), 64) == VF_OK, "an empty corpus is a valid file");
    // ---- random mutations of valid headers
    std::mt19937_64 rng(12345);
    int accepted = 0;
    for (int it = 0; it < 4000; ++it) {
        Header h = good((uint32_t)(rng() % 3), rng() % 200, 1 + (uint32_t)(rng() % 64), (uint32_t)(rng() & 1));
        const size_t pay = (size_t)(h.n * h.d * esz(h.dtype) + ((h.flags & 1) ? h.n * 8 : 0));
        unsigned char* raw = (unsigned char*)&h;
        const int flips = 1 + (int)(rng() % 3);
        for (int f = 0; f < flips; ++f) {
            const int at = (int)(rng() % 32);                 // magic, version, dtype, n, d, flags
            switch (rng() % 4) {
                case 0: raw[at] ^= (unsigned char)(1u << (rng() % 8)); break;
                case 1: raw[at] = (unsigned char)rng(); break;
                case 2: raw[at] = 0xff; break;
                default: raw[at] = 0; break;
            }
        }
        const size_t cut = (rng() % 4 == 0) ? (size_t)(rng() % (pay + 1)) : pay;
        const std::string p = write_file("mut.vfc", h, cut);
        accepted += probe(p, 64 + (long long)cut) == VF_OK;
    }
    printf("mutated headers: %d of 4000 still describe a file the parser accepts\n", accepted);
    // ---- argument checks in front of the device code
    vf_index* ix = (vf_index*)0x1;
    float x[8] = {0}; int64_t ids[4] = {0}; float sc[4] = {0}; int32_t dev[1] = {0}; int32_t many[65] = {0};
    CHECK(vf_index_create(nullptr, x, 1, 4, 0, 0, 0) == VF_EINVAL, "null out");
    CHECK(vf_index_create(&ix, nullptr, 3, 4, 0, 0, 0) == VF_EINVAL && ix == nullptr, "null rows");
    CHECK(vf_index_create(&ix, x, -1, 4, 0, 0, 0) == VF_EINVAL, "negative n");
    CHECK(vf_index_create(&ix, x, 2, 0, 0, 0, 0) == VF_EINVAL, "d = 0");
    CHECK(vf_index_create(&ix, x, 2, 4, 9, 0, 0) == VF_EINVAL, "dtype 9");
    CHECK(vf_index_create(&ix, x, (int64_t)1 << 33, 4, 0, 0, 0) != VF_OK, "2^33 rows from an 8-float buffer is refused before it is read");
    CHECK(vf_index_create_sharded(&ix, x, 2, 4, 0, nullptr, 1) == VF_EINVAL, "null device list");
    CHECK(vf_index_create_sharded(&ix, x, 2, 4, 0, dev, 0) == VF_EINVAL, "zero devices");
    CHECK(vf_index_create_sharded(&ix, x, 2, 4, 0, many, 65) == VF_EINVAL, "65 devices");
    CHECK(vf_index_create_sharded(nullptr, x, 2, 4, 0, dev, 1) == VF_EINVAL, "null out (sharded)");
    CHECK(vf_index_group(&ix, nullptr, 2) == VF_EINVAL, "null shard list");
    CHECK(vf_index_group(&ix, &ix, 0) == VF_EINVAL, "zero shards");
    CHECK(vf_index_search(nullptr, x, 1, 1, ids, sc) == VF_EINVAL, "null handle");
    CHECK(vf_index_search_end(nullptr, 0) == VF_EINVAL, "null handle (end)");
    CHECK(vf_index_set_option(nullptr, "wide", 1) == VF_EINVAL, "null handle (option)");
    CHECK(vf_index_stats(nullptr, nullptr) == VF_EINVAL, "null handle (stats)");
    CHECK(vf_index_shards(nullptr, nullptr, nullptr, 0) == VF_EINVAL, "null handle (shards)");
    CHECK(vf_index_peer_access(nullptr, nullptr, 0, nullptr) == VF_EINVAL, "null handle (peer)");
    CHECK(vf_cosine_matrix_rows(nullptr, ids, 2, sc) == VF_EINVAL, "null handle (rows)");
    CHECK(vf_cosine_scores(x, -1, x, 1, 4, sc, 0) == VF_EINVAL, "negative na");
    CHECK(vf_cosine_scores(x, 1, x, 1, 0, sc, 0) == VF_EINVAL, "d = 0 (cosine)");
    CHECK(vf_cosine_scores(x, 0, x, 5, 4, sc, 0) == VF_OK, "empty product is a no-op");
    CHECK(vf_cosine_scores(nullptr, 1, x, 1, 4, sc, 0) == VF_EINVAL, "null a");
    CHECK(vf_fuse_rank(x, x, -1, sc, ids, 0) == VF_EINVAL && vf_fuse_rank(x, x, 4097, sc, ids, 0) == VF_EINVAL, "fuse_rank sizes");
    CHECK(vf_fuse_rank(nullptr, x, 2, sc, ids, 0) == VF_EINVAL, "fuse_rank null");
    CHECK(vf_merge_topk_device(nullptr, nullptr, 0, 1, 1, ids, sc, 0, nullptr) == VF_EINVAL, "merge sizes");
    CHECK(vf_merge_topk_packed_device(nullptr, 2, 1, 1, ids, sc, 0, nullptr) == VF_EINVAL, "merge null");
    CHECK(vf_merge_topk_packed_device(x, 200, 1, 1000, ids, sc, 0, nullptr) == VF_EUNSUPPORTED, "merge nparts x k");
    int32_t ndev = -5;
    (void)vf_device_count(&ndev);
    CHECK(ndev >= 0, "device count");
    CHECK(vf_device_count(nullptr) == VF_EINVAL, "null count");
    printf("host sanitize check: %d failure(s); devices visible: %d\n", g_fail, ndev);
    return g_fail ? 1 : 0;
}
