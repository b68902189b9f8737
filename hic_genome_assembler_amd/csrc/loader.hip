// loader.hip - host-side HiC-Pro triplet parser (no device code): the "next" row N1 of SURVEY.md 8f.
//
// Replaces buildAdjacencyMatrix's line loop (scaffoldToChromosomes.py:70-98, orderGenome.py:65-93):
//   id1 <TAB> id2 <TAB> value  ->  dense fp64 matrix, each triplet mirrored, unknown bin IDs skipped,
//   a cell named twice keeps the value of the LATER line.
// The reference fills a Python list of lists (about 32 bytes per cell, minutes of interpreter time at
// N >= 16k).  Here the file is mmap'ed, cut into per-thread chunks at line boundaries and parsed with
// a correctly rounded decimal-to-double conversion (like Python's float()): Clinger's exact fast path
// for <= 15 significant digits - HiC-Pro prints 6 decimals - and glibc's strtod_l otherwise.  To keep "later line wins" exact under
// parallelism, parsed triplets are bucketed by the row block that owns the destination cell and each
// block is written by one thread walking the buckets in file order.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <clocale>
#include <cstdlib>
#include <locale.h>
#include <cstdint>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/hicmi.h"

namespace hicmi { int set_error(int code, const char* msg); }   // api.hip: records the message for hicmi_last_error()

namespace {
struct Cell { int32_t r, c; double v; };

struct ParseError { bool bad = false; int64_t offset = 0; };

const double kPow10[23] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16,
                           1e17, 1e18, 1e19, 1e20, 1e21, 1e22};

// Correctly rounded double from the decimal text [p, end).  Fast path (W. Clinger, 1990): with at most
// 15 significant digits the integer significand and 10^|e| (|e| <= 22) are both exact doubles, so one
// multiplication or division rounds once - the correctly rounded result.  Anything else (long
// significands, big exponents, inf/nan) goes to strtod_l in the "C" locale.
inline bool parse_double(const char* p, const char* end, const char* file_end, locale_t cloc, double& out)
{
    const char* q = p;
    bool neg = false;
    if (q < end && (*q == '-' || *q == '+')) { neg = *q == '-'; q++; }
    uint64_t w = 0; int digits = 0, dec_exp = 0; bool any = false, fast = true;
    while (q < end && *q >= '0' && *q <= '9') { any = true; if (w || *q != '0') { if (digits < 19) { w = w * 10 + (uint64_t)(*q - '0'); digits++; } else fast = false; } q++; }
    if (q < end && *q == '.') {
        q++;
        while (q < end && *q >= '0' && *q <= '9') {
            any = true;
            if (w || *q != '0') { if (digits < 19) { w = w * 10 + (uint64_t)(*q - '0'); digits++; dec_exp--; } else fast = false; }
            else dec_exp--;
            q++;
        }
    }
    if (any && q < end && (*q == 'e' || *q == 'E')) {
        const char* r = q + 1;
        bool eneg = false;
        if (r < end && (*r == '-' || *r == '+')) { eneg = *r == '-'; r++; }
        if (r < end && *r >= '0' && *r <= '9') {
            int ev = 0;
            while (r < end && *r >= '0' && *r <= '9') { if (ev < 10000) ev = ev * 10 + (*r - '0'); r++; }
            dec_exp += eneg ? -ev : ev;
            q = r;
        }
    }
    if (any && q == end && fast && digits <= 15 && dec_exp >= -22 && dec_exp <= 22) {
        double v = (double)w;
        v = dec_exp < 0 ? v / kPow10[-dec_exp] : v * kPow10[dec_exp];
        out = neg ? -v : v;
        return true;
    }
    // general case: strtod_l needs a terminator after the token; '\t', '\r', '\n' do that inside the
    // mapping, only a token that touches the end of the file is copied out
    char buf[512];
    const char* src = p;
    if (end == file_end) {
        size_t len = (size_t)(end - p);
        if (len >= sizeof(buf)) return false;
        memcpy(buf, p, len); buf[len] = 0;
        src = buf;
    }
    char* stop = nullptr;
    double v = strtod_l(src, &stop, cloc);
    if (stop != src + (end - p) || stop == src) return false;
    if (end - p > 1 && (src[0] == '0' || ((src[0] == '-' || src[0] == '+') && src[1] == '0')) &&
        (src[1] == 'x' || src[1] == 'X' || src[2] == 'x' || src[2] == 'X')) return false;     // float() has no hex form
    out = v;
    return true;
}

inline const char* parse_i64(const char* p, const char* end, int64_t& out, bool& ok)
{
    bool neg = false;
    if (p < end && (*p == '-' || *p == '+')) { neg = *p == '-'; p++; }
    if (p >= end || *p < '0' || *p > '9') { ok = false; return p; }
    int64_t v = 0;
    while (p < end && *p >= '0' && *p <= '9') { v = v * 10 + (*p - '0'); p++; }
    out = neg ? -v : v;
    return p;
}
}  // namespace

extern "C" int hicmi_load_hicpro_matrix(const char* path, const int64_t* bin_ids, int64_t n, double* out, int threads,
                                        int64_t* edges_out)
{
    auto hicmi_set_error = [](int code, const char* msg) { return hicmi::set_error(code, msg); };
    if (!path || !bin_ids || !out || n < 0) return hicmi_set_error(HICMI_EINVAL, "bad arguments");
    if (edges_out) *edges_out = 0;
    const bool prof = getenv("HICMI_LOADER_PROFILE") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    auto t_start = now();
    auto t_zero = now();
    int fd = open(path, O_RDONLY);
    if (fd < 0) return hicmi_set_error(HICMI_EINVAL, (std::string("cannot open ") + path).c_str());
    struct stat st;
    if (fstat(fd, &st) != 0) { close(fd); return hicmi_set_error(HICMI_EINVAL, "fstat failed"); }
    const size_t size = (size_t)st.st_size;
    if (size == 0 || n == 0) { close(fd); std::memset(out, 0, sizeof(double) * (size_t)n * (size_t)n); return HICMI_OK; }
    const char* data = (const char*)mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (data == MAP_FAILED) return hicmi_set_error(HICMI_ENOMEM, "mmap failed");
    madvise((void*)data, size, MADV_SEQUENTIAL);

    int64_t max_id = -1;
    for (int64_t i = 0; i < n; i++) {
        if (bin_ids[i] < 0) { munmap((void*)data, size); return hicmi_set_error(HICMI_EINVAL, "negative bin ID"); }
        max_id = std::max(max_id, bin_ids[i]);
    }
    std::vector<int32_t> lookup((size_t)max_id + 2, -1);
    for (int64_t i = 0; i < n; i++) lookup[(size_t)bin_ids[i]] = (int32_t)i;      // a repeated ID keeps its last row, like a dict

    int T = threads > 0 ? threads : (int)std::thread::hardware_concurrency();
    T = std::max(1, std::min(T, 64));
    if (size < (size_t)T * 4096) T = 1;
    // chunk boundaries at line starts
    std::vector<size_t> cut((size_t)T + 1, size);
    cut[0] = 0;
    for (int t = 1; t < T; t++) {
        size_t p = size / (size_t)T * (size_t)t;
        while (p < size && data[p] != '\n') p++;
        cut[(size_t)t] = p < size ? p + 1 : size;
    }
    const int64_t rows_per = (n + T - 1) / T;
    std::vector<std::vector<std::vector<Cell>>> bucket((size_t)T, std::vector<std::vector<Cell>>((size_t)T));
    std::vector<ParseError> errs((size_t)T);
    std::vector<int64_t> edges((size_t)T, 0);

    locale_t cloc = newlocale(LC_ALL_MASK, "C", (locale_t)0);
    auto parse = [&](int t) {
        const char* p = data + cut[(size_t)t];
        const char* end = data + cut[(size_t)t + 1];
        auto& mine = bucket[(size_t)t];
        for (auto& b : mine) b.reserve((size_t)((end - p) / 24 / T + 16));
        while (p < end) {
            const char* eol = (const char*)memchr(p, '\n', (size_t)(end - p));
            if (!eol) eol = end;
            const char* le = eol;
            while (le > p && le[-1] == '\r') le--;                 // the reference strips '\r' and '\n' (S2C:82)
            if (le > p) {
                bool ok = true;
                int64_t a = 0, b = 0;
                const char* q = parse_i64(p, le, a, ok);
                if (ok && q < le && *q == '\t') q = parse_i64(q + 1, le, b, ok); else ok = false;
                double v = 0.0;
                if (ok && q < le && *q == '\t') {
                    q++;
                    while (q < le && (*q == ' ')) q++;             // float() tolerates surrounding blanks
                    const char* vend = (const char*)memchr(q, '\t', (size_t)(le - q));   // further columns are ignored (cols[2])
                    if (!vend) vend = le;
                    while (vend > q && vend[-1] == ' ') vend--;
                    if (q >= vend || !parse_double(q, vend, data + size, cloc, v)) ok = false;
                } else ok = false;
                if (!ok) { errs[(size_t)t].bad = true; errs[(size_t)t].offset = (int64_t)(p - data); return; }
                if (a >= 0 && a <= max_id && b >= 0 && b <= max_id) {
                    const int32_t ia = lookup[(size_t)a], ib = lookup[(size_t)b];
                    if (ia >= 0 && ib >= 0) {
                        mine[(size_t)(ia / rows_per)].push_back({ia, ib, v});
                        mine[(size_t)(ib / rows_per)].push_back({ib, ia, v});
                        edges[(size_t)t]++;
                    }
                }
            } else {
                // an empty line makes the reference raise (int('') at S2C:83); refuse it as well
                errs[(size_t)t].bad = true; errs[(size_t)t].offset = (int64_t)(p - data); return;
            }
            p = eol + 1;
        }
    };
    {
        std::vector<std::thread> pool;
        for (int t = 0; t < T; t++) pool.emplace_back(parse, t);
        for (auto& th : pool) th.join();
    }
    auto t_parse = now();
    if (cloc) freelocale(cloc);
    for (int t = 0; t < T; t++)
        if (errs[(size_t)t].bad) {
            munmap((void*)data, size);
            return hicmi_set_error(HICMI_EINVAL, (std::string("malformed triplet line at byte ") + std::to_string(errs[(size_t)t].offset)).c_str());
        }
    auto apply = [&](int owner) {
        // each owner zero-fills its own row block first: the page faults of the n*n output run in parallel
        const int64_t r0 = std::min<int64_t>(n, (int64_t)owner * rows_per), r1 = std::min<int64_t>(n, r0 + rows_per);
        if (r1 > r0) std::memset(out + (size_t)r0 * (size_t)n, 0, sizeof(double) * (size_t)(r1 - r0) * (size_t)n);
        for (int src = 0; src < T; src++)                          // file order: chunk by chunk, line by line
            for (const Cell& c : bucket[(size_t)src][(size_t)owner]) out[(size_t)c.r * (size_t)n + (size_t)c.c] = c.v;
    };
    {
        std::vector<std::thread> pool;
        for (int t = 0; t < T; t++) pool.emplace_back(apply, t);
        for (auto& th : pool) th.join();
    }
    auto t_apply = now();
    munmap((void*)data, size);
    if (prof) fprintf(stderr, "[hicmi] loader: %d threads, zero %.0f ms, parse %.0f ms, apply %.0f ms\n", T, ms(t_start, t_zero),
                      ms(t_zero, t_parse), ms(t_parse, t_apply));
    int64_t total = 0;
    for (int t = 0; t < T; t++) total += edges[(size_t)t];
    if (edges_out) *edges_out = total;
    return HICMI_OK;
}

// ---- valid-pair scan (orientSmallScaffolds.py:159-177 readValidPairFile) ---------
// HiC-Pro allValidPairs lines: read <TAB> scaffold1 <TAB> pos1 <TAB> strand1 <TAB> scaffold2 <TAB> pos2 ...
// The reference keeps, for every ORDERED pair of neighbouring scaffold names it registered, the two positions of
// each line naming that pair; everything else in a multi-GB file is skipped.  Same host-side scheme as the
// matrix loader: mmap, per-thread chunks cut at line starts, hits collected per thread and concatenated in
// thread order = file order.
#include <unordered_map>
#include <string_view>

namespace {
struct PairHit { int32_t pair; int64_t p1, p2; };
struct ScanResult { std::vector<PairHit> hits; };

inline bool parse_pos(const char* p, const char* end, int64_t& out)
{
    // Python int(): optional sign, digits, surrounding blanks tolerated
    while (p < end && (*p == ' ')) p++;
    while (end > p && (end[-1] == ' ')) end--;
    bool ok = true;
    const char* q = parse_i64(p, end, out, ok);
    return ok && q == end;
}
}  // namespace

extern "C" int hicmi_scan_valid_pairs(const char* path, const char* names_blob, const int64_t* name_off, int64_t n_names,
                                      const int32_t* pair_a, const int32_t* pair_b, int64_t n_pairs, int threads,
                                      int64_t* n_hits_out, int64_t* n_lines_out, void** handle_out)
{
    auto err = [](int code, const std::string& msg) { return hicmi::set_error(code, msg.c_str()); };
    if (!path || !names_blob || !name_off || !n_hits_out || !n_lines_out || !handle_out || n_names < 0 || n_pairs < 0 ||
        (n_pairs > 0 && (!pair_a || !pair_b)))
        return err(HICMI_EINVAL, "bad arguments");
    *n_hits_out = 0; *n_lines_out = 0; *handle_out = nullptr;
    std::unordered_map<std::string_view, int32_t> name_id;
    name_id.reserve((size_t)n_names * 2);
    for (int64_t i = 0; i < n_names; i++)
        name_id.emplace(std::string_view(names_blob + name_off[i], (size_t)(name_off[i + 1] - name_off[i])), (int32_t)i);
    std::unordered_map<uint64_t, int32_t> pair_id;
    pair_id.reserve((size_t)n_pairs * 2);
    for (int64_t i = 0; i < n_pairs; i++) {
        if (pair_a[i] < 0 || pair_a[i] >= n_names || pair_b[i] < 0 || pair_b[i] >= n_names) return err(HICMI_EINVAL, "pair names out of range");
        pair_id[((uint64_t)(uint32_t)pair_a[i] << 32) | (uint32_t)pair_b[i]] = (int32_t)i;
    }
    int fd = open(path, O_RDONLY);
    if (fd < 0) return err(HICMI_EINVAL, std::string("cannot open ") + path);
    struct stat st;
    if (fstat(fd, &st) != 0) { close(fd); return err(HICMI_EINVAL, "fstat failed"); }
    const size_t size = (size_t)st.st_size;
    auto* res = new ScanResult();
    if (size == 0) { close(fd); *handle_out = res; return HICMI_OK; }
    const char* data = (const char*)mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (data == MAP_FAILED) { delete res; return err(HICMI_ENOMEM, "mmap failed"); }
    madvise((void*)data, size, MADV_SEQUENTIAL);
    int T = threads > 0 ? threads : (int)std::thread::hardware_concurrency();
    T = std::max(1, std::min(T, 64));
    if (size < (size_t)T * 4096) T = 1;
    std::vector<size_t> cut((size_t)T + 1, size);
    cut[0] = 0;
    for (int t = 1; t < T; t++) {
        size_t p = size / (size_t)T * (size_t)t;
        while (p < size && data[p] != '\n') p++;
        cut[(size_t)t] = p < size ? p + 1 : size;
    }
    std::vector<std::vector<PairHit>> part((size_t)T);
    std::vector<int64_t> lines((size_t)T, 0), bad((size_t)T, -1);
    auto work = [&](int t) {
        const char* p = data + cut[(size_t)t];
        const char* end = data + cut[(size_t)t + 1];
        auto& out = part[(size_t)t];
        while (p < end) {
            const char* eol = (const char*)memchr(p, '\n', (size_t)(end - p));
            if (!eol) eol = end;
            const char* le = eol;
            while (le > p && le[-1] == '\r') le--;
            // columns 0..5
            const char* col[7];
            int nc = 0;
            col[nc++] = p;
            for (const char* q = p; nc < 7;) {
                const char* tab = (const char*)memchr(q, '\t', (size_t)(le - q));
                if (!tab) break;
                col[nc++] = tab + 1;
                q = tab + 1;
            }
            // the reference indexes cols[1] and cols[4] on every line (orientSmallScaffolds.py:168) and cols[2], cols[5]
            // only for a registered pair (:170): a 5-column line that names no such pair is accepted there, so it is here
            if (nc < 5) { bad[(size_t)t] = (int64_t)(p - data); return; }   // cols[4] missing: IndexError in the reference
            const char* c4_end = nc >= 6 ? col[5] - 1 : le;
            const char* c5_end = nc >= 7 ? col[6] - 1 : le;
            lines[(size_t)t]++;
            auto a = name_id.find(std::string_view(col[1], (size_t)(col[2] - 1 - col[1])));
            if (a != name_id.end()) {
                auto b = name_id.find(std::string_view(col[4], (size_t)(c4_end - col[4])));
                if (b != name_id.end()) {
                    auto hit = pair_id.find(((uint64_t)(uint32_t)a->second << 32) | (uint32_t)b->second);
                    if (hit != pair_id.end()) {
                        if (nc < 6) { bad[(size_t)t] = (int64_t)(p - data); return; }   // cols[5] missing on a hit: IndexError there too
                        int64_t p1 = 0, p2 = 0;
                        if (!parse_pos(col[2], col[3] - 1, p1) || !parse_pos(col[5], c5_end, p2)) { bad[(size_t)t] = (int64_t)(p - data); return; }
                        out.push_back({hit->second, p1, p2});
                    }
                }
            }
            p = eol + 1;
        }
    };
    {
        std::vector<std::thread> pool;
        for (int t = 0; t < T; t++) pool.emplace_back(work, t);
        for (auto& th : pool) th.join();
    }
    munmap((void*)data, size);
    for (int t = 0; t < T; t++)
        if (bad[(size_t)t] >= 0) { delete res; return err(HICMI_EINVAL, "malformed valid-pair line at byte " + std::to_string(bad[(size_t)t])); }
    size_t total = 0;
    for (auto& v : part) total += v.size();
    res->hits.reserve(total);
    for (auto& v : part) res->hits.insert(res->hits.end(), v.begin(), v.end());
    int64_t n_lines = 0;
    for (auto v : lines) n_lines += v;
    *n_hits_out = (int64_t)total; *n_lines_out = n_lines; *handle_out = res;
    return HICMI_OK;
}

extern "C" int hicmi_scan_fetch(void* handle, int32_t* pair_idx, int64_t* pos1, int64_t* pos2)
{
    if (!handle) return hicmi::set_error(HICMI_EINVAL, "NULL handle");
    auto* res = reinterpret_cast<ScanResult*>(handle);
    if (!res->hits.empty() && (!pair_idx || !pos1 || !pos2)) return hicmi::set_error(HICMI_EINVAL, "NULL output");
    for (size_t i = 0; i < res->hits.size(); i++) { pair_idx[i] = res->hits[i].pair; pos1[i] = res->hits[i].p1; pos2[i] = res->hits[i].p2; }
    delete res;
    return HICMI_OK;
}
