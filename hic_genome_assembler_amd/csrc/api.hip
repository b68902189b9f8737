// api.hip - the C ABI of libhicmi.so (include/hicmi.h): context, device buffers, stage drivers,
// and the small host-side pieces of SciPy's linkage post-processing.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <unordered_map>
#include <functional>
#include <chrono>
#include <vector>

#include "../../include/hicmi.h"
#include "hicmi_internal.h"
#include "hyper.h"

using namespace hicmi;

static thread_local std::string g_err;

static int fail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

namespace hicmi { int set_error(int code, const char* msg) { return fail(code, "%s", msg); } }

#define HIPCHK(expr)                                                                              \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) return fail(HICMI_EHIP, "%s: %s", #expr, hipGetErrorString(_e));     \
    } while (0)

enum Family { F_ROW_SUMS, F_BUILD_W, F_NNCHAIN, F_SORT, F_RANK_INVERT, F_CUT_COUNT, F_HYPER_FLAGS, F_P2_SELECT,
              F_P2_TOTAL, F_P2_SCORE, F_P2_EXACT, F_P2_INSERT, F_P2_WINDOW_G, F_P2_WINDOW_DELTA, F_PLOT, F_PRESORT, F_RANK_RELABEL,
              F_RANK_TIED, F_P2_WINDOW_FLOPS, F_COUNT };
static const char* kFamilyNames[F_COUNT] = {"row_sums", "build_w", "nnchain", "sort_rows", "rank_invert",
                                            "cut_count", "hyper_flags", "p2_select", "p2_total", "p2_score",
                                            "p2_score_exact", "p2_score_insert", "p2_window_G", "p2_window_delta", "plot",
                                            "presort_rows", "rank_relabel", "rank_rows_tied",
                                            "p2_window_G_flops"};      // (its "bytes" are FLOPS of the window tables' GEMM)

constexpr int kBaseSlabs = 256;                        // partial sums of the closed-form BASE term (one slab per workgroup)

struct TimedRegion { int fam; hipEvent_t a, b; };

struct hicmi_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    // contacts
    int64_t n = 0, ldc = 0;
    double* dC = nullptr;
    bool own_c = false;
    double *d_np = nullptr, *d_seq = nullptr;
    bool have_sums = false;
    // row shard (one map over several GPUs): this context sorts / counts only rows first, first + stride, ...
    int64_t shard_first = 0, shard_stride = 1;
    // upgma
    double *dW = nullptr, *dW2 = nullptr; int64_t ldw = 0; int64_t w_rows = 0;
    int *d_size = nullptr, *d_chain = nullptr, *d_status = nullptr;
    double* d_zraw = nullptr;
    std::vector<double> zraw;
    // rank matrix
    int32_t* d_order = nullptr;
    uint16_t *dR = nullptr, *dRank = nullptr; int64_t ldr = 0; int64_t r_rows = 0;
    void* d_sort_scratch = nullptr; size_t sort_scratch_cap = 0;
    bool have_rank = false;
    // pre-sort: the rank rows in storage labels, computed on a second stream while the nn-chain runs (hicmi_upgma)
    hipStream_t stream2 = nullptr; hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    uint16_t* dRankS = nullptr; int64_t rank_s_rows = 0;
    int32_t* d_ident = nullptr; int64_t ident_cap = 0;
    unsigned char* d_ties = nullptr; int64_t ties_cap = 0;   // [count of flagged rows, 16 bytes][one flag per storage row]
    uint16_t* d_tie_bits = nullptr; int64_t tie_bits_rows = 0, ld_bits = 0;   // per storage row: "same key as the element before"
    int32_t* d_row_list = nullptr; int64_t row_list_cap = 0;
    int64_t scan_first_batch = 0, scan_last_improved = 0;   // hicmi_p2_scan_all: the first batch size of the next round
    int64_t presort_tied_rows = 0;                        // ... and how many rows it re-sorted because they hold equal keys
    int presort_used = 0;                                 // last hicmi_rank_matrix: 0 sorted itself, 1 relabelled the pre-sort, 2 pre-sort discarded (ties)
    int64_t presort_n = 0;                                // > 0: dRankS holds the rows of the current n x n matrix
    bool presort_dealt = false;                           // its rows were dealt out by a counter (workgroups on the chain's XCD left)
    // cut scan
    int32_t* d_x = nullptr; uint8_t* d_sig = nullptr; int64_t x_cap = 0;
    unsigned char* d_scan_prog = nullptr; int64_t scan_prog_cap = 0;     // device-driven scan loops: state record + lists
    int64_t cached_start = -1;
    double* d_tmp = nullptr; int64_t tmp_cap = 0;
    // part 2
    double* dM2 = nullptr; int64_t n2 = 0, ld2 = 0, m2_cap = 0;
    int32_t* d_sel = nullptr; int64_t sel_cap = 0;
    double* d_H = nullptr; int64_t h_cap = 0;
    int32_t* d_perms = nullptr; int64_t perms_cap = 0;
    double* d_scores = nullptr; int64_t scores_cap = 0;
    double* d_partial = nullptr; int64_t partial_cap = 0;
    double* d_T = nullptr; int64_t t_cap = 0;
    // part 2 search state: layout (scaffold ranges of the selection), arrangement, window tables
    int32_t *d_scaf_start = nullptr, *d_scaf_len = nullptr; int64_t scaf_cap = 0, n_scaf = 0;
    std::vector<int32_t> h_scaf_start, h_scaf_len;
    int32_t* d_arr_packed = nullptr; int64_t arr_cap = 0;       // [S ids][S+1 positions][S reversed flags]
    std::vector<int32_t> h_arr_packed;
    std::vector<int32_t> h_arr_id, h_arr_pos; std::vector<uint8_t> h_arr_rev;
    int32_t* d_pos2sel = nullptr; int64_t pos_cap = 0; int64_t n_arr = 0;
    int8_t* d_orders = nullptr; uint8_t* d_orients = nullptr; int64_t ord_cap = 0, ori_cap = 0;
    int tab_k = 0; int64_t n_orders = 0, n_orients = 0;
    std::vector<int8_t> h_orders; std::vector<uint8_t> h_orients;
    std::vector<int32_t> h_pos2sel;                              // host mirror of the arrangement's bin order
    uint64_t arr_version = 0;                                    // bumped by every hicmi_p2_set_arrangement
    uint64_t cur_lit_version = ~0ull; double cur_lit_total = 0.0, cur_lit_value = 0.0;   // literal score of the arrangement itself
    double cache_total = 0.0; bool cache_valid = false;          // literal scores under one total, keyed by bin order
    std::unordered_map<std::string, double> exact_cache;
    double* d_G = nullptr; int64_t g_cap = 0;
    double* d_delta = nullptr; int64_t delta_cap = 0;
    WindowBatchEntry* d_wb = nullptr; int64_t wb_cap = 0;
    // device-decided insertion (k_part2_insert.hip): second arrangement buffers (ping-pong) and work areas
    int32_t* d_arr_packed2 = nullptr; int64_t arr2_cap = 0;
    int32_t* d_pos2sel2 = nullptr; int64_t pos2_cap = 0;
    double* d_ins_T = nullptr; int64_t ins_t_cap = 0;
    double* d_ins_partial = nullptr; int64_t ins_partial_cap = 0;
    unsigned char* d_ins_blob = nullptr; int64_t ins_blob_cap = 0;   // per job: [InsState][InsLog x steps]
    InsStep* d_ins_steps = nullptr; int64_t ins_steps_cap = 0;       // [step][job] records of a lock-step queue
    // plot support
    int32_t* d_plot_order = nullptr; int64_t plot_order_cap = 0;
    unsigned char* d_plot_work = nullptr; int64_t plot_work_cap = 0;
    double* d_plot_img = nullptr; int64_t plot_img_cap = 0;
    // pinned staging: pageable hipMemcpyAsync takes a slow, serialising path in the runtime, which hurts when
    // several contexts are driven from different host threads
    char* pin_up = nullptr; size_t pin_up_cap = 0, pin_up_off = 0;
    char* pin_down = nullptr; size_t pin_down_cap = 0;

    // timing
    int timing = 0;                                       // 0 off, 1 every family, 2 only the families launched a few times per map
    std::vector<TimedRegion> regions;
    std::vector<hipEvent_t> pool;
    double ms[F_COUNT] = {0}; int64_t launches[F_COUNT] = {0}; double bytes[F_COUNT] = {0};
    // nn-chain counters since the last hicmi_timing_reset (reported by hicmi_nnchain_stats)
    double nn_scans = 0, nn_scan_cols = 0, nn_cache_hits = 0, nn_merges = 0; int64_t nn_retries = 0;
};

namespace {
hipError_t sync_stream(hicmi_ctx* c);

struct Timed {
    hicmi_ctx* c; int fam; hipEvent_t a = nullptr, b = nullptr; hipStream_t st;
    Timed(hicmi_ctx* ctx, int f, double algo_bytes, hipStream_t on_stream = nullptr) : c(ctx), fam(f), st(on_stream ? on_stream : ctx->stream)
    {
        c->launches[f]++; c->bytes[f] += algo_bytes;
        if (!on()) return;
        a = grab(); b = grab();
        hipEventRecord(a, st);
    }
    // Event pairs around the hundreds of small launches of the scans and of Part 2 cost about 10 ms per 16k map;
    // mode 2 keeps them for the families that are launched a handful of times (the dominant kernel is one of them).
    bool on() const { return c->timing == 1 || (c->timing == 2 && (fam <= F_RANK_INVERT || fam >= F_PRESORT)); }
    ~Timed()
    {
        if (!on()) return;
        hipEventRecord(b, st);
        c->regions.push_back({fam, a, b});
    }
    hipEvent_t grab()
    {
        if (!c->pool.empty()) { hipEvent_t e = c->pool.back(); c->pool.pop_back(); return e; }
        hipEvent_t e; hipEventCreate(&e); return e;
    }
};

int resolve_timing(hicmi_ctx* c)
{
    if (c->regions.empty()) return HICMI_OK;
    HIPCHK(sync_stream(c));
    if (c->stream2) HIPCHK(hipStreamSynchronize(c->stream2));
    for (auto& r : c->regions) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) c->ms[r.fam] += ms;
        c->pool.push_back(r.a); c->pool.push_back(r.b);
    }
    c->regions.clear();
    return HICMI_OK;
}


hipError_t sync_stream(hicmi_ctx* c)
{
    hipError_t e = hipStreamSynchronize(c->stream);
    c->pin_up_off = 0;                                   // everything staged for upload has been consumed
    return e;
}

// copy `bytes` from pageable host memory to the device through the pinned upload arena (asynchronous)
int upload(hicmi_ctx* c, void* dst, const void* src, size_t bytes)
{
    if (bytes == 0) return HICMI_OK;
    const size_t need = (bytes + 63) & ~(size_t)63;
    if (c->pin_up_off + need > c->pin_up_cap) {
        HIPCHK(sync_stream(c));
        if (need > c->pin_up_cap) {
            if (c->pin_up) (void)hipHostFree(c->pin_up);
            c->pin_up = nullptr; c->pin_up_cap = 0;
            size_t cap = std::max<size_t>(need * 2, (size_t)1 << 20);
            HIPCHK(hipHostMalloc((void**)&c->pin_up, cap, hipHostMallocDefault));
            c->pin_up_cap = cap;
        }
    }
    char* slot = c->pin_up + c->pin_up_off;
    memcpy(slot, src, bytes);
    c->pin_up_off += need;
    HIPCHK(hipMemcpyAsync(dst, slot, bytes, hipMemcpyHostToDevice, c->stream));
    return HICMI_OK;
}

int ensure_pin_down(hicmi_ctx* c, size_t bytes)
{
    if (bytes > c->pin_down_cap) {
        if (c->pin_down) (void)hipHostFree(c->pin_down);
        c->pin_down = nullptr; c->pin_down_cap = 0;
        size_t cap = std::max<size_t>(bytes * 2, (size_t)1 << 20);
        HIPCHK(hipHostMalloc((void**)&c->pin_down, cap, hipHostMallocDefault));
        c->pin_down_cap = cap;
    }
    return HICMI_OK;
}

// device -> pageable host through the pinned download buffer; synchronises the stream
int download(hicmi_ctx* c, void* dst, const void* src, size_t bytes)
{
    if (bytes == 0) { HIPCHK(sync_stream(c)); return HICMI_OK; }
    int rc_pin = ensure_pin_down(c, bytes);
    if (rc_pin) return rc_pin;
    HIPCHK(hipMemcpyAsync(c->pin_down, src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(sync_stream(c));
    memcpy(dst, c->pin_down, bytes);
    return HICMI_OK;
}

template <typename T>
int ensure(T*& p, int64_t& cap, int64_t need)
{
    if (need <= cap && p) return HICMI_OK;
    if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
    if (need <= 0) need = 1;
    hipError_t e = hipMalloc((void**)&p, (size_t)need * sizeof(T));
    if (e != hipSuccess) { p = nullptr; return fail(HICMI_ENOMEM, "hipMalloc(%lld bytes): %s", (long long)(need * (int64_t)sizeof(T)), hipGetErrorString(e)); }
    cap = need;
    return HICMI_OK;
}

void free_dev(void* p) { if (p) (void)hipFree(p); }

void drop_matrix_state(hicmi_ctx* c)
{
    if (c->own_c) free_dev(c->dC);
    c->dC = nullptr; c->own_c = false; c->n = 0; c->ldc = 0;
    free_dev(c->d_np); free_dev(c->d_seq); c->d_np = c->d_seq = nullptr; c->have_sums = false;
    c->have_rank = false; c->cached_start = -1; c->n2 = 0;
    if (c->presort_n && c->stream2) (void)hipStreamSynchronize(c->stream2);    // the pre-sort reads the matrix being dropped
    c->presort_n = 0;
}

int alloc_sums(hicmi_ctx* c)
{
    HIPCHK(hipMalloc((void**)&c->d_np, sizeof(double) * (size_t)std::max<int64_t>(c->n, 1)));
    HIPCHK(hipMalloc((void**)&c->d_seq, sizeof(double) * (size_t)std::max<int64_t>(c->n, 1)));
    return HICMI_OK;
}

int compute_sums(hicmi_ctx* c)
{
    if (c->have_sums) return HICMI_OK;
    if (!c->dC) return fail(HICMI_EINVAL, "no contact matrix set");
    {
        Timed t(c, F_ROW_SUMS, 2.0 * 8.0 * (double)c->n * (double)c->n);
        launch_row_sums(c->dC, c->ldc, (int)c->n, c->d_np, c->d_seq, 0, 1, c->stream);
    }
    HIPCHK(hipGetLastError());
    c->have_sums = true;
    return HICMI_OK;
}
}  // namespace

extern "C" {

int hicmi_abi_version(void) { return HICMI_ABI_VERSION; }
const char* hicmi_last_error(void) { return g_err.c_str(); }

int hicmi_device_count(int* count)
{
    if (!count) return fail(HICMI_EINVAL, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; return fail(HICMI_EHIP, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
    *count = n;
    return HICMI_OK;
}

int hicmi_create(int device, hicmi_ctx** out)
{
    if (!out) return fail(HICMI_EINVAL, "out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(HICMI_EHIP, "no HIP device available (%s): libhicmi has no CPU fallback",
                    e != hipSuccess ? hipGetErrorString(e) : "device count 0");
    if (device < 0 || device >= n) return fail(HICMI_EINVAL, "device %d out of range (0..%d)", device, n - 1);
    HIPCHK(hipSetDevice(device));
    hicmi_ctx* c = new hicmi_ctx();
    c->device = device;
    hipError_t se = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (se != hipSuccess) { delete c; return fail(HICMI_EHIP, "hipStreamCreate: %s", hipGetErrorString(se)); }
    *out = c;
    return HICMI_OK;
}

int hicmi_destroy(hicmi_ctx* c)
{
    if (!c) return HICMI_OK;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    drop_matrix_state(c);
    free_dev(c->dW); free_dev(c->dW2); free_dev(c->d_size); free_dev(c->d_chain); free_dev(c->d_status); free_dev(c->d_zraw);
    free_dev(c->d_order); free_dev(c->dR); free_dev(c->dRank); free_dev(c->d_sort_scratch);
    free_dev(c->dRankS); free_dev(c->d_ident); free_dev(c->d_ties); free_dev(c->d_row_list); free_dev(c->d_tie_bits);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->stream2) (void)hipStreamDestroy(c->stream2);
    free_dev(c->d_x); free_dev(c->d_sig); free_dev(c->d_tmp); free_dev(c->d_scan_prog);
    free_dev(c->dM2); free_dev(c->d_sel); free_dev(c->d_H); free_dev(c->d_perms); free_dev(c->d_scores);
    free_dev(c->d_partial); free_dev(c->d_T);
    free_dev(c->d_scaf_start); free_dev(c->d_scaf_len); free_dev(c->d_arr_packed);
    free_dev(c->d_pos2sel); free_dev(c->d_orders); free_dev(c->d_orients);
    free_dev(c->d_G); free_dev(c->d_delta); free_dev(c->d_wb);
    free_dev(c->d_arr_packed2); free_dev(c->d_pos2sel2); free_dev(c->d_ins_T); free_dev(c->d_ins_partial);
    free_dev(c->d_ins_blob); free_dev(c->d_ins_steps);
    free_dev(c->d_plot_order); free_dev(c->d_plot_work); free_dev(c->d_plot_img);
    if (c->pin_up) (void)hipHostFree(c->pin_up);
    if (c->pin_down) (void)hipHostFree(c->pin_down);
    for (auto& r : c->regions) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    for (auto e : c->pool) (void)hipEventDestroy(e);
    (void)hipStreamDestroy(c->stream);
    delete c;
    return HICMI_OK;
}

int hicmi_stream(hicmi_ctx* c, void** stream_out)
{
    if (!c || !stream_out) return fail(HICMI_EINVAL, "NULL argument");
    *stream_out = (void*)c->stream;
    return HICMI_OK;
}

int hicmi_synchronize(hicmi_ctx* c)
{
    if (!c) return fail(HICMI_EINVAL, "NULL context");
    HIPCHK(sync_stream(c));
    return HICMI_OK;
}

// ---------------------------------------------------------------------------------------------------
int hicmi_set_contacts_host(hicmi_ctx* c, const double* contacts, int64_t n)
{
    if (!c || !contacts || n < 1) return fail(HICMI_EINVAL, "bad arguments");
    if (n > 65536) return fail(HICMI_EUNSUPPORTED, "n = %lld > 65536 bins: rank matrix is uint16 in this version", (long long)n);
    HIPCHK(hipSetDevice(c->device));
    drop_matrix_state(c);
    c->n = n; c->ldc = n;
    HIPCHK(hipMalloc((void**)&c->dC, sizeof(double) * (size_t)n * (size_t)n));
    c->own_c = true;
    HIPCHK(hipMemcpyAsync(c->dC, contacts, sizeof(double) * (size_t)n * (size_t)n, hipMemcpyHostToDevice, c->stream));
    HIPCHK(sync_stream(c));
    return alloc_sums(c);
}

int hicmi_set_contacts_host_f32(hicmi_ctx* c, const float* contacts, int64_t n)
{
    // BASELINE configs[4]: a 64,000-bin map stored as fp32 (16.4 GB instead of 32.8 GB on the host and over PCIe).
    // The values are widened on the device; every stage computes in fp64 on exactly those widened values.
    if (!c || !contacts || n < 1) return fail(HICMI_EINVAL, "bad arguments");
    if (n > 65536) return fail(HICMI_EUNSUPPORTED, "n = %lld > 65536 bins: rank matrix is uint16 in this version", (long long)n);
    HIPCHK(hipSetDevice(c->device));
    drop_matrix_state(c);
    c->n = n; c->ldc = n;
    const size_t cells = (size_t)n * (size_t)n;
    HIPCHK(hipMalloc((void**)&c->dC, sizeof(double) * cells));
    c->own_c = true;
    // the fp32 image is staged in the tail of the fp64 buffer and widened back to front, row block by row block
    float* d_stage = reinterpret_cast<float*>(c->dC) + cells;
    HIPCHK(hipMemcpyAsync(d_stage, contacts, sizeof(float) * cells, hipMemcpyHostToDevice, c->stream));
    launch_widen_f32(d_stage, c->dC, (int64_t)cells, c->stream);
    HIPCHK(hipGetLastError());
    HIPCHK(sync_stream(c));
    return alloc_sums(c);
}

int hicmi_set_contacts_device(hicmi_ctx* c, const double* d_contacts, int64_t n, int64_t ld)
{
    if (!c || !d_contacts || n < 1 || ld < n) return fail(HICMI_EINVAL, "bad arguments");
    if (n > 65536) return fail(HICMI_EUNSUPPORTED, "n = %lld > 65536 bins: rank matrix is uint16 in this version", (long long)n);
    HIPCHK(hipSetDevice(c->device));
    drop_matrix_state(c);
    c->n = n; c->ldc = ld; c->dC = const_cast<double*>(d_contacts); c->own_c = false;
    return alloc_sums(c);
}

int hicmi_contacts_device(hicmi_ctx* c, void** d_contacts_out, int64_t* n_out, int64_t* ld_out)
{
    if (!c || !d_contacts_out || !n_out || !ld_out) return fail(HICMI_EINVAL, "NULL argument");
    if (!c->dC) return fail(HICMI_EINVAL, "no contact matrix set");
    *d_contacts_out = (void*)c->dC; *n_out = c->n; *ld_out = c->ldc;
    return HICMI_OK;
}

int hicmi_set_row_shard(hicmi_ctx* c, int64_t first, int64_t stride)
{
    if (!c || stride < 1 || first < 0 || first >= stride) return fail(HICMI_EINVAL, "row shard needs 0 <= first < stride");
    c->shard_first = first; c->shard_stride = stride;
    c->have_rank = false; c->cached_start = -1;
    if (c->presort_n && c->stream2) (void)hipStreamSynchronize(c->stream2);
    c->presort_n = 0;
    return HICMI_OK;
}

int hicmi_set_row_sums(hicmi_ctx* c, const double* np_sum, const double* seq_sum)
{
    if (!c || !np_sum || !seq_sum) return fail(HICMI_EINVAL, "bad arguments");
    if (!c->dC) return fail(HICMI_EINVAL, "no contact matrix set");
    HIPCHK(hipSetDevice(c->device));
    int rc = upload(c, c->d_np, np_sum, sizeof(double) * (size_t)c->n);
    if (rc) return rc;
    rc = upload(c, c->d_seq, seq_sum, sizeof(double) * (size_t)c->n);
    if (rc) return rc;
    HIPCHK(sync_stream(c));
    if (c->presort_n && c->stream2) HIPCHK(hipStreamSynchronize(c->stream2));
    c->presort_n = 0;                                      // the similarity keys depend on the sums
    c->have_sums = true;
    return HICMI_OK;
}

int hicmi_row_sums(hicmi_ctx* c, double* np_sum, double* seq_sum)
{
    if (!c) return fail(HICMI_EINVAL, "NULL context");
    HIPCHK(hipSetDevice(c->device));
    int rc = HICMI_OK;
    if (c->shard_stride > 1 && !c->have_sums) {
        // this shard's rows only (the other entries read 0); the caller gathers the shards and hands the complete
        // vectors back with hicmi_set_row_sums
        if (!c->dC) return fail(HICMI_EINVAL, "no contact matrix set");
        HIPCHK(hipMemsetAsync(c->d_np, 0, sizeof(double) * (size_t)c->n, c->stream));
        HIPCHK(hipMemsetAsync(c->d_seq, 0, sizeof(double) * (size_t)c->n, c->stream));
        {
            Timed t(c, F_ROW_SUMS, 2.0 * 8.0 * (double)c->n * (double)c->n / (double)c->shard_stride);
            launch_row_sums(c->dC, c->ldc, (int)c->n, c->d_np, c->d_seq, (int)c->shard_first, (int)c->shard_stride, c->stream);
        }
        HIPCHK(hipGetLastError());
    }
    else rc = compute_sums(c);
    if (rc) return rc;
    if (np_sum) { rc = download(c, np_sum, c->d_np, sizeof(double) * (size_t)c->n); if (rc) return rc; }
    if (seq_sum) { rc = download(c, seq_sum, c->d_seq, sizeof(double) * (size_t)c->n); if (rc) return rc; }
    if (!np_sum && !seq_sum) HIPCHK(sync_stream(c));
    return HICMI_OK;
}

int hicmi_compact(hicmi_ctx* c, const int32_t* keep, int64_t n_keep)
{
    if (!c || !keep || n_keep < 1) return fail(HICMI_EINVAL, "bad arguments");
    if (!c->dC) return fail(HICMI_EINVAL, "no contact matrix set");
    if (n_keep > c->n) return fail(HICMI_EINVAL, "n_keep > n");
    for (int64_t i = 0; i < n_keep; i++)
        if (keep[i] < 0 || keep[i] >= c->n || (i && keep[i] <= keep[i - 1])) return fail(HICMI_EINVAL, "keep must be ascending indices in [0, n)");
    HIPCHK(hipSetDevice(c->device));
    int32_t* d_keep = nullptr; double* d_new = nullptr;
    struct Guard { int32_t*& a; double*& b; ~Guard() { free_dev(a); free_dev(b); } } guard{d_keep, d_new};   // error paths
    HIPCHK(hipMalloc((void**)&d_keep, sizeof(int32_t) * (size_t)n_keep));
    HIPCHK(hipMalloc((void**)&d_new, sizeof(double) * (size_t)n_keep * (size_t)n_keep));
    {
        int rc_up = upload(c, d_keep, keep, sizeof(int32_t) * (size_t)n_keep);
        if (rc_up) return rc_up;
    }
    launch_compact(c->dC, c->ldc, d_keep, (int)n_keep, d_new, n_keep, c->stream);
    HIPCHK(hipGetLastError());
    HIPCHK(sync_stream(c));
    free_dev(d_keep); d_keep = nullptr;
    double* kept = d_new; d_new = nullptr;                     // ownership moves to the context below
    drop_matrix_state(c);
    c->dC = kept; c->own_c = true; c->n = n_keep; c->ldc = n_keep;
    int rc = alloc_sums(c);
    if (rc) return rc;
    return compute_sums(c);
}

int hicmi_selftest_division(hicmi_ctx* c, uint64_t seed, int64_t samples, uint64_t* mismatches_out)
{
    if (!c || !mismatches_out || samples < 1) return fail(HICMI_EINVAL, "bad arguments");
    HIPCHK(hipSetDevice(c->device));
    unsigned long long* d = nullptr;
    HIPCHK(hipMalloc((void**)&d, sizeof(unsigned long long)));
    HIPCHK(hipMemsetAsync(d, 0, sizeof(unsigned long long), c->stream));
    const int blocks = 2048, iters = (int)std::max<int64_t>(1, samples / (blocks * 256));
    launch_selftest_division(seed, blocks, iters, d, c->stream);
    HIPCHK(hipGetLastError());
    unsigned long long bad = 0;
    HIPCHK(hipMemcpyAsync(&bad, d, sizeof(bad), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(sync_stream(c));
    (void)hipFree(d);
    *mismatches_out = bad;
    return HICMI_OK;
}

// ---------------------------------------------------------------------------------------------------
int hicmi_label_linkage(const double* zraw, int64_t n, double* Z)
{
    if (!zraw || !Z || n < 1) return fail(HICMI_EINVAL, "bad arguments");
    const int64_t m = n - 1;
    // numpy argsort(kind='mergesort') on the heights: stable.  (height, index) pairs side by side: the comparator of an
    // index sort fetched two scattered heights per comparison - 1.1 ms at 16k bins, 6.3 ms at 64k
    std::vector<std::pair<double, int64_t>> keyed((size_t)m);
    for (int64_t i = 0; i < m; i++) keyed[(size_t)i] = {zraw[4 * i + 2], i};
    std::stable_sort(keyed.begin(), keyed.end(),
                     [](const std::pair<double, int64_t>& x, const std::pair<double, int64_t>& y) { return x.first < y.first; });
    std::vector<int64_t> idx((size_t)m);
    for (int64_t i = 0; i < m; i++) idx[(size_t)i] = keyed[(size_t)i].second;
    std::vector<int64_t> parent((size_t)(2 * n - 1)), sz((size_t)(2 * n - 1), 0);
    for (int64_t i = 0; i < 2 * n - 1; i++) { parent[i] = i; if (i < n) sz[i] = 1; }
    auto find = [&](int64_t x) {
        int64_t p = x;
        while (parent[x] != x) x = parent[x];
        while (parent[p] != x) { int64_t nx = parent[p]; parent[p] = x; p = nx; }
        return x;
    };
    int64_t next = n;
    for (int64_t r = 0; r < m; r++) {
        const double* s = zraw + 4 * idx[r];
        int64_t a = find((int64_t)s[0]), b = find((int64_t)s[1]);
        Z[4 * r + 0] = (double)std::min(a, b);
        Z[4 * r + 1] = (double)std::max(a, b);
        Z[4 * r + 2] = s[2];
        parent[a] = next; parent[b] = next;
        sz[next] = sz[a] + sz[b];
        Z[4 * r + 3] = (double)sz[next];
        next++;
    }
    return HICMI_OK;
}

int hicmi_leaf_order(const double* Z, int64_t n, int32_t* leaves)
{
    if (!Z || !leaves || n < 1) return fail(HICMI_EINVAL, "bad arguments");
    if (n == 1) { leaves[0] = 0; return HICMI_OK; }
    std::vector<int64_t> stack;
    stack.reserve(64);
    stack.push_back(2 * n - 2);
    int64_t out = 0;
    while (!stack.empty()) {
        int64_t node = stack.back(); stack.pop_back();
        if (node < n) { if (out >= n) return fail(HICMI_ESTATE, "malformed linkage"); leaves[out++] = (int32_t)node; continue; }
        const double* row = Z + 4 * (node - n);
        int64_t aa = (int64_t)row[0], ab = (int64_t)row[1];
        int64_t na = aa < n ? 1 : (int64_t)Z[4 * (aa - n) + 3];
        int64_t nb = ab < n ? 1 : (int64_t)Z[4 * (ab - n) + 3];
        // count_sort='ascending': smaller child first; on equal counts keep (Z[i,0], Z[i,1]) order
        if (na > nb) { stack.push_back(aa); stack.push_back(ab); }
        else         { stack.push_back(ab); stack.push_back(aa); }
    }
    return out == n ? HICMI_OK : fail(HICMI_ESTATE, "malformed linkage (%lld leaves of %lld)", (long long)out, (long long)n);
}

// Buffers of the rank matrix: dRank (result), d_order ([order][inverse]), dR (argsort rows of the bitonic path), scratch.
static int ensure_rank_buffers(hicmi_ctx* c, int64_t n, int64_t ldr, bool bitonic)
{
    if (c->r_rows < n || c->ldr != ldr || !c->dRank) {
        free_dev(c->dR); free_dev(c->dRank); free_dev(c->d_order); c->dR = c->dRank = nullptr; c->d_order = nullptr;
        HIPCHK(hipMalloc((void**)&c->dRank, sizeof(uint16_t) * (size_t)n * (size_t)ldr));
        HIPCHK(hipMalloc((void**)&c->d_order, sizeof(int32_t) * 2 * (size_t)n));       // [order][inverse order]
        c->ldr = ldr; c->r_rows = n;
    }
    if (bitonic && !c->dR) HIPCHK(hipMalloc((void**)&c->dR, sizeof(uint16_t) * (size_t)n * (size_t)ldr));
    size_t need = bitonic ? sort_scratch_bytes((int)n) : sort_radix_scratch_bytes((int)n);
    if (need > c->sort_scratch_cap) {
        free_dev(c->d_sort_scratch); c->d_sort_scratch = nullptr; c->sort_scratch_cap = 0;
        HIPCHK(hipMalloc(&c->d_sort_scratch, need));
        c->sort_scratch_cap = need;
    }
    return HICMI_OK;
}

// The pre-sort.  The nn-chain keeps 8 of the 256 CUs busy for 60 % of Part 1, and the row sort that follows it needs the
// chain's result only for two things: the numbering of rows and columns (the leaf order) and the order of EQUAL
// similarities inside a row.  A row without equal similarities has the same sorted sequence under any numbering, so its
// rank row in leaf labels is the rank row in storage labels re-addressed: rank[a][b] = rank_s[order[a]][order[b]].
// While the chain runs on the main stream, a second low-priority stream therefore sorts every row in storage labels
// (same kernel, identity order, on at most 192 CUs so that the chain's workgroups and its flush kernels always find a
// free one) and notes which rows hold equal keys, and where; hicmi_rank_matrix then only relabels (k_rank_relabel, ~0.5 ms
// at 16k), and finishes the rows with equal keys (sparse maps, fp32 contacts: possibly all of them) with a sort of
// 32-bit (run, leaf position) keys - k_sort_tied.hip.  HICMI_NO_PRESORT=1 disables.
static int start_presort(hicmi_ctx* c)
{
    const bool off = getenv("HICMI_NO_PRESORT") != nullptr || getenv("HICMI_SORT_RADIX") != nullptr
                     || getenv("HICMI_SORT_LDS") != nullptr;
    const char* from = getenv("HICMI_PRESORT_FROM");               // (tests lower it; below ~2000 bins the sort is 0.3 ms)
    const int64_t n = c->n;
    c->presort_used = 0;
    if (off || n < (from ? atoll(from) : 2048)) return HICMI_OK;     // (a row shard pre-sorts ALL rows: it is idle meanwhile)
    if (c->presort_n == n) return HICMI_OK;                       // same matrix, same sums: still valid
    const int64_t ldr = (n + 63) & ~(int64_t)63;
    int rc = ensure_rank_buffers(c, n, ldr, true);
    if (rc) return rc;
    if (!c->stream2) {
        int lo = 0, hi = 0;
        HIPCHK(hipDeviceGetStreamPriorityRange(&lo, &hi));         // lo = least urgent
        HIPCHK(hipStreamCreateWithPriority(&c->stream2, hipStreamNonBlocking, lo));
    }
    if (!c->ev_fork) {
        HIPCHK(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
    }
    if (c->ties_cap < n) {
        free_dev(c->d_ties); c->d_ties = nullptr;
        HIPCHK(hipMalloc((void**)&c->d_ties, 16 + (size_t)n));
        c->ties_cap = n;
    }
    const int64_t ld_bits = std::max<int64_t>(sort_padded_size((int)n), 16) / 16;
    if (c->tie_bits_rows < n || c->ld_bits != ld_bits) {
        free_dev(c->d_tie_bits); c->d_tie_bits = nullptr;
        HIPCHK(hipMalloc((void**)&c->d_tie_bits, sizeof(uint16_t) * (size_t)n * (size_t)ld_bits));
        c->tie_bits_rows = n; c->ld_bits = ld_bits;
    }
    if (c->rank_s_rows < n || !c->dRankS) {
        free_dev(c->dRankS); c->dRankS = nullptr;
        HIPCHK(hipMalloc((void**)&c->dRankS, sizeof(uint16_t) * (size_t)n * (size_t)ldr));
        c->rank_s_rows = n;
    }
    if (c->ident_cap < n) {
        free_dev(c->d_ident); c->d_ident = nullptr;
        HIPCHK(hipMalloc((void**)&c->d_ident, sizeof(int32_t) * (size_t)n));
        std::vector<int32_t> id((size_t)n);
        for (int64_t i = 0; i < n; i++) id[(size_t)i] = (int32_t)i;
        int rc_up = upload(c, c->d_ident, id.data(), sizeof(int32_t) * (size_t)n);
        if (rc_up) return rc_up;
        c->ident_cap = n;
    }
    HIPCHK(hipEventRecord(c->ev_fork, c->stream));                // sums and identity are in place
    HIPCHK(hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
    HIPCHK(hipMemsetAsync(c->d_ties, 0, 16 + (size_t)n, c->stream2));
    {
        Timed t(c, F_PRESORT, (8.0 + 2.0) * (double)n * (double)n, c->stream2);      // N^2 (e + idx) for the row argsort
        SortExtras x;
        x.tie_count = reinterpret_cast<unsigned*>(c->d_ties); x.tie_flag = c->d_ties + 16;
        x.tie_limit = (unsigned)n;                                 // (never gives up: tied rows are finished by k_rank_rows_tied)
        x.tie_bits = c->d_tie_bits; x.ld_bits = ld_bits;
        x.max_workgroups = 192;                                    // (the chain: up to 64 single-wave workgroups, one CU each)
        x.avoid_xcc = nnchain_local_xcc((int)n);                   // ... all on one XCD, which this kernel's workgroups leave alone
        x.row_counter = reinterpret_cast<unsigned*>(c->d_ties) + 1;     // (zeroed with the tie count just above)
        if (getenv("HICMI_TEST_PRESORT_ALL_LEAVE")) x.avoid_xcc = -2;      // test hook: every workgroup behaves as if it were on that XCD
        c->presort_dealt = x.avoid_xcc >= 0 || x.avoid_xcc == -2;
        launch_sort_rows(c->dC, c->ldc, c->d_ident, c->d_ident, c->d_np, c->d_seq, (int)n, c->d_sort_scratch, c->dR, ldr, 0, 1,
                         c->stream2, x);
    }
    {
        // (its own family: with the chain's parties holding the LDS of one XCD's CUs, the eighth of this kernel's workgroups
        //  that is dealt to that XCD - 32 KB of LDS each at 16k - waits for the end of a chain epoch; 0.3 ms of work that
        //  can read as 8 ms, all of it beside the chain)
        Timed t(c, F_RANK_INVERT, (2.0 + 2.0) * (double)n * (double)n, c->stream2);
        launch_rank_invert(c->dR, c->dRankS, ldr, (int)n, 0, 1, c->stream2);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(c->ev_join, c->stream2));
    c->presort_n = n;
    c->have_rank = false;                                          // dR is being rewritten
    return HICMI_OK;
}

int hicmi_upgma(hicmi_ctx* c, double* Z_out, int32_t* leaves_out)
{
    if (!c || !leaves_out) return fail(HICMI_EINVAL, "bad arguments");
    if (!c->dC) return fail(HICMI_EINVAL, "no contact matrix set");
    HIPCHK(hipSetDevice(c->device));
    int rc = compute_sums(c);
    if (rc) return rc;
    const int64_t n = c->n;
    if (n == 1) { leaves_out[0] = 0; c->zraw.clear(); return HICMI_OK; }
    const int64_t ldw = (n + 15) & ~(int64_t)15;
    if (c->w_rows < n || c->ldw != ldw || !c->dW) {
        free_dev(c->dW); free_dev(c->dW2); c->dW = c->dW2 = nullptr;
        HIPCHK(hipMalloc((void**)&c->dW, sizeof(double) * (size_t)n * (size_t)ldw));
        HIPCHK(hipMalloc((void**)&c->dW2, sizeof(double) * (size_t)n * (size_t)ldw));
        c->ldw = ldw; c->w_rows = n;
        free_dev(c->d_size); free_dev(c->d_chain); free_dev(c->d_zraw); free_dev(c->d_status);
        c->d_size = c->d_chain = c->d_status = nullptr; c->d_zraw = nullptr;
        HIPCHK(hipMalloc((void**)&c->d_size, nnchain_workspace_bytes((int)n)));
        HIPCHK(hipMalloc((void**)&c->d_chain, sizeof(int) * 64 * (size_t)(n + 2)));     // one copy per workgroup of k_nn_epoch_w1 (64) / _mw / _mwc (16)
        HIPCHK(hipMalloc((void**)&c->d_zraw, sizeof(double) * 4 * (size_t)n));
        HIPCHK(hipMalloc((void**)&c->d_status, sizeof(int)));
    }
    static const bool step_prof = getenv("HICMI_STEP_PROFILE") != nullptr;
    std::vector<std::pair<const char*, std::chrono::steady_clock::time_point>> marks;
    auto mark = [&](const char* what) { if (step_prof) marks.emplace_back(what, std::chrono::steady_clock::now()); };
    mark("start");
    {
        Timed t(c, F_BUILD_W, 8.0 * (0.5 * (double)n * (double)n + (double)n * (double)n));
        launch_build_w(c->dC, c->ldc, c->d_np, (int)n, c->dW, ldw, c->stream);
    }
    HIPCHK(hipGetLastError());
    // The pre-sort runs beside the chain on the second stream - from the moment the chain's FIRST cache pass (k_nn_rowmin over
    // the whole matrix, the one full-chip pass the chain has) is queued: started right after k_build_w the pre-sort's 192
    // big workgroups shared the CUs with that pass and stretched it from 0.5 to 2.4 ms at 16k, from 2 to 9.8 ms at 32k.
    int presort_rc = HICMI_OK;
    bool presort_started = false;
    auto presort_now = [&] { if (!presort_started) { presort_started = true; presort_rc = start_presort(c); } };
    mark("build_w queued");
    // The nn-chain.  Algorithmic bytes: 8 B x (sum over row scans of the live columns + 3 x sum over merges
    // of the live columns), with the scans counted by the kernels themselves (a scan the neighbour cache answers moves
    // nothing); the merge term is 3 * 8 * sum_{k=0}^{n-2} (n - k).
    const bool prof_on = getenv("HICMI_NNCHAIN_PROFILE") != nullptr;
    // merges between two column flushes = merges per epoch launch.  Every row a merge reads is patched at the columns
    // whose writes are still deferred, so a merge's cost grows with the list: 1024 -> 256 merges per epoch took the chain
    // from 124.8 to 118.0 ms at 16k and from 282.5 to 272.4 ms at 32k (128: no further gain - ~40 us of launches per epoch)
    const char* cap = getenv("HICMI_NNCHAIN_DCAP");
    struct { int state[16]; unsigned long long prof[8]; unsigned char mail[1152]; unsigned long long detail[32]; } nn;   // the head of the workspace
    for (int attempt = 0; attempt < 3; attempt++) {
        {
            Timed t(c, F_NNCHAIN, 0.0);
            int epochs = launch_nnchain(c->dW, c->dW2, ldw, (int)n, c->d_chain, c->d_zraw, c->d_size, prof_on,
                                        cap ? atoi(cap) : 256, getenv("HICMI_NNCHAIN_NO_COMPACT") == nullptr, attempt, c->stream,
                                        presort_now);
            presort_now();                                         // (whatever path the chain took)
            if (presort_rc) return presort_rc;
            // the family is reported per epoch launch (the flush / compaction launches in between are ~1 % of it)
            if (epochs > 1) c->launches[F_NNCHAIN] += epochs - 1;
        }
        HIPCHK(hipGetLastError());
        int rc_dl = download(c, &nn, nnchain_state_ptr(c->d_size), sizeof(nn));
        if (rc_dl) return rc_dl;
        c->bytes[F_NNCHAIN] += 8.0 * ((double)nn.prof[5] + 3.0 * (0.5 * (double)(n - 1) * (double)(n + 2)));
        c->nn_scans += (double)nn.prof[6]; c->nn_scan_cols += (double)nn.prof[5]; c->nn_cache_hits += (double)nn.prof[7];
        c->nn_merges += (double)(n - 1);
        if (nn.state[5] != 2 || attempt > 1) break;
        // A peer workgroup of the column-sliced chain did not answer within its spin budget (a GPU shared with other
        // work can delay a workgroup's start): nothing is wrong with the data.  Rebuild the distances and run the whole
        // chain again - spread over the chip first (had its parties asked for one XCD), then on one workgroup, which
        // waits for nobody.
        fprintf(stderr, "[hicmi] nn-chain: a peer workgroup answered late at merge %d; re-running %s\n", nn.state[0],
                attempt == 0 ? "with the parties spread over the chip" : "on one workgroup");
        c->nn_retries++;
        launch_build_w(c->dC, c->ldc, c->d_np, (int)n, c->dW, ldw, c->stream);
        HIPCHK(hipGetLastError());
    }
    mark("chain (epochs, compactions, state)");
    c->zraw.assign((size_t)(4 * (n - 1)), 0.0);
    {
        int rc_dl = download(c, c->zraw.data(), c->d_zraw, sizeof(double) * 4 * (size_t)(n - 1));
        if (rc_dl) return rc_dl;
    }
    mark("merge records to the host");
    if (prof_on) {
        const unsigned long long* pr = nn.prof;
        fprintf(stderr, "[hicmi] nnchain phases (ms @100MHz): bookkeeping %.2f scan %.2f pick %.2f merge %.2f update %.2f; "
                        "%llu scans (%.2f per merge), %llu cached steps\n",
                pr[0] / 1e5, pr[1] / 1e5, pr[2] / 1e5, pr[3] / 1e5, pr[4] / 1e5, pr[6], (double)pr[6] / (double)(n - 1), pr[7]);
        const unsigned long long* dq = nn.detail;
        if (dq[0] | dq[2] | dq[3])
            fprintf(stderr, "[hicmi] fused pass detail (ms): issue gathers %.2f, LDS pass %.2f, loads arrive %.2f, compute+stores %.2f, "
                            "gathered+reductions %.2f, stores acked %.2f, barrier %.2f\n",
                    dq[0] / 1e5, dq[1] / 1e5, dq[2] / 1e5, dq[3] / 1e5, dq[4] / 1e5, dq[5] / 1e5, dq[6] / 1e5);
        if (dq[9])
            fprintf(stderr, "[hicmi] one-wave kernel: %llu exchanges, post + polling %.2f ms, %.2f polls of the slowest peer per exchange\n",
                    dq[9], dq[8] / 1e5, (double)dq[10] / (double)dq[9]);
        if (dq[9])
            fprintf(stderr, "[hicmi] stand-alone scans by cause: chain restart %llu, exact tie %llu, neighbour of the merged cluster %llu, "
                            "new neighbour of the row scanned with the merge %llu, reached over cached steps %llu, other %llu\n",
                    dq[12], dq[13], dq[14], dq[15], dq[16], dq[17]);
    }
    if (nn.state[5] == 2)
        return fail(HICMI_ESTATE, "nn-chain: a peer workgroup of the column-sliced kernel answered late, and so did the retry");
    if (nn.state[5] == 3)
        return fail(HICMI_ESTATE, "nn-chain: the replicas of the column-sliced kernel disagree about a merge (stale read between "
                                  "workgroups); set HICMI_NNCHAIN_WGS=1 to run on one workgroup");
    if (nn.state[5] != 0 || nn.state[0] != (int)(n - 1))
        return fail(HICMI_ESTATE, "nn-chain kernel stopped on its guard at merge %d of %lld (NaN distances or an internal error; check %d)",
                    nn.state[0], (long long)(n - 1), nn.state[13]);
    std::vector<double> Z((size_t)(4 * (n - 1)));
    rc = hicmi_label_linkage(c->zraw.data(), n, Z.data());
    if (rc) return rc;
    mark("sort + label");
    rc = hicmi_leaf_order(Z.data(), n, leaves_out);
    if (rc) return rc;
    if (Z_out) memcpy(Z_out, Z.data(), sizeof(double) * Z.size());
    mark("leaf order");
    if (step_prof) {
        fprintf(stderr, "[hicmi] upgma host timeline (ms):");
        for (size_t i = 1; i < marks.size(); i++)
            fprintf(stderr, " %s %.2f%s", marks[i].first, std::chrono::duration<double, std::milli>(marks[i].second - marks[i - 1].second).count(),
                    i + 1 < marks.size() ? "," : "\n");
    }
    return HICMI_OK;
}

int hicmi_get_raw_merges(hicmi_ctx* c, double* out)
{
    if (!c || !out) return fail(HICMI_EINVAL, "bad arguments");
    if (c->zraw.empty()) return fail(HICMI_EINVAL, "hicmi_upgma has not run");
    memcpy(out, c->zraw.data(), sizeof(double) * c->zraw.size());
    return HICMI_OK;
}

int hicmi_nnchain_stats(hicmi_ctx* c, double* out6)
{
    if (!c || !out6) return fail(HICMI_EINVAL, "bad arguments");
    out6[0] = c->nn_merges; out6[1] = c->nn_scans; out6[2] = c->nn_scan_cols; out6[3] = c->nn_cache_hits;
    out6[4] = (double)c->nn_retries; out6[5] = 0.0;
    return HICMI_OK;
}

// ---------------------------------------------------------------------------------------------------
int hicmi_rank_matrix(hicmi_ctx* c, const int32_t* order)
{
    if (!c || !order) return fail(HICMI_EINVAL, "bad arguments");
    if (!c->dC) return fail(HICMI_EINVAL, "no contact matrix set");
    HIPCHK(hipSetDevice(c->device));
    const int64_t n = c->n;
    static const bool tprof = getenv("HICMI_STEP_PROFILE") != nullptr;
    std::vector<std::chrono::steady_clock::time_point> tm;
    auto tick = [&]() { if (tprof) tm.push_back(std::chrono::steady_clock::now()); };
    auto report = [&]() {
        if (!tprof || tm.size() < 2) return;
        fprintf(stderr, "[hicmi] rank_matrix host timeline (ms):");
        for (size_t i = 1; i < tm.size(); i++) fprintf(stderr, " %.2f", std::chrono::duration<double, std::milli>(tm[i] - tm[i - 1]).count());
        fprintf(stderr, "\n");
    };
    tick();
    std::vector<uint8_t> seen((size_t)n, 0);
    for (int64_t i = 0; i < n; i++) {
        if (order[i] < 0 || order[i] >= n || seen[(size_t)order[i]]) return fail(HICMI_EINVAL, "order is not a permutation of 0..n-1");
        seen[(size_t)order[i]] = 1;
    }
    int rc = compute_sums(c);
    if (rc) return rc;
    const int64_t ldr = (n + 63) & ~(int64_t)63;
    // Default: the register-blocked bitonic network (argsort rows R, then the inversion pass).  HICMI_SORT_RADIX=1: the LSD
    // radix kernel, which writes the rank rows directly - built to get rid of the O(n log^2 n) stages and of the scratch
    // round trips of long rows, verified against the same tests, and measured SLOWER on MI355X (16k / 32k / 64k bins:
    // 15.7 / 87 / 581 ms against 11.7 / 50.6 / 236 ms): 18 passes x 7 workgroup barriers and random 4-byte LDS scatters
    // cost more than the 105 register-resident stages.  Kept as the second implementation the large-map tests compare with.
    const bool bitonic = getenv("HICMI_SORT_RADIX") == nullptr;
    if (c->presort_n != n) c->presort_used = 0;
    tick();
    if (c->presort_n && c->stream2) HIPCHK(hipStreamSynchronize(c->stream2));   // (the buffers below are the pre-sort's too)
    tick();
    rc = ensure_rank_buffers(c, n, ldr, bitonic);
    if (rc) return rc;
    {
        std::vector<int32_t> both((size_t)(2 * n));
        for (int64_t i = 0; i < n; i++) { both[(size_t)i] = order[i]; both[(size_t)(n + order[i])] = (int32_t)i; }
        int rc_up = upload(c, c->d_order, both.data(), sizeof(int32_t) * both.size());
        if (rc_up) return rc_up;
    }
    const double share = 1.0 / (double)c->shard_stride;             // this shard's rows only
    if (c->presort_n == n && bitonic && c->presort_dealt) {
        // The pre-sort's workgroups that were dealt to the chain's XCD left at once and the others took the rows from a
        // counter.  Which workgroup runs where is the hardware's business: had ALL of them landed on that XCD, no row
        // would have been sorted - the counter says how many were handed out.
        unsigned head[4] = {0, 0, 0, 0};
        int rc_dl = download(c, head, c->d_ties, sizeof(head));
        if (rc_dl) return rc_dl;
        if ((int64_t)head[1] < n) {
            fprintf(stderr, "[hicmi] pre-sort: only %u of %lld rows were handed out; sorting now\n", head[1], (long long)n);
            c->presort_n = 0; c->presort_used = 0;
        }
    }
    if (c->presort_n == n && bitonic) {
        // rows sorted in storage labels while the nn-chain ran (start_presort): re-addressed by the leaf order; rows that
        // hold equal keys get the order inside their runs from k_rank_rows_tied
        tick();
        std::vector<unsigned char> ties(16 + (size_t)n);
        int rc_dl = download(c, ties.data(), c->d_ties, ties.size());
        if (rc_dl) return rc_dl;
        tick();
        unsigned n_tied = 0;
        memcpy(&n_tied, ties.data(), sizeof(n_tied));
        // HICMI_PRESORT_TIES=resort: the rows with equal keys go through the full 64-bit sort again (the first version of
        // this path, kept as the A/B for k_rank_rows_tied); it gives the pre-sort up when more than half of the rows do
        const bool resort = getenv("HICMI_PRESORT_TIES") != nullptr && !strcmp(getenv("HICMI_PRESORT_TIES"), "resort");
        c->presort_used = (resort && n_tied > (unsigned)(n / 2)) ? 2 : 1;
        c->presort_tied_rows = (int64_t)n_tied;
        if (c->presort_used == 1) {
            // rows (in leaf numbering) whose storage row holds equal keys: the order inside their runs of equal keys
            // depends on the leaf labels
            std::vector<int32_t> again;                            // (of this shard's rows)
            const int64_t own = (n - c->shard_first + c->shard_stride - 1) / c->shard_stride;
            for (int64_t a = c->shard_first; a < n && n_tied; a += c->shard_stride)
                if (ties[16 + (size_t)order[a]]) again.push_back((int32_t)a);
            if ((int64_t)again.size() < own) {
                Timed t(c, F_RANK_RELABEL, (2.0 + 2.0) * (double)n * (double)(own - (int64_t)again.size()));
                launch_rank_relabel(c->dRankS, c->dRank, ldr, (int)n, c->d_order, (int)c->shard_first, (int)c->shard_stride,
                                    c->stream);
            }
            HIPCHK(hipGetLastError());
            if (!again.empty()) {
                if (c->row_list_cap < n) {
                    free_dev(c->d_row_list); c->d_row_list = nullptr;
                    HIPCHK(hipMalloc((void**)&c->d_row_list, sizeof(int32_t) * 2 * (size_t)n));     // the list + one flag per listed row
                    c->row_list_cap = n;
                }
                int rc_up = upload(c, c->d_row_list, again.data(), sizeof(int32_t) * again.size());
                if (rc_up) return rc_up;
                const double part = (double)again.size() / (double)n;
                if (resort) {
                    c->presort_n = 0;                              // dR (the storage-label argsort rows) is overwritten
                    SortExtras x;
                    x.row_list = c->d_row_list; x.n_list = (int)again.size();
                    {
                        Timed t(c, F_SORT, (8.0 + 2.0) * (double)n * (double)n * part);
                        launch_sort_rows(c->dC, c->ldc, c->d_order, c->d_order + n, c->d_np, c->d_seq, (int)n, c->d_sort_scratch,
                                         c->dR, ldr, 0, 1, c->stream, x);
                    }
                    HIPCHK(hipGetLastError());
                    Timed t(c, F_RANK_INVERT, (2.0 + 2.0) * (double)n * (double)n * part);
                    launch_rank_invert(c->dR, c->dRank, ldr, (int)n, 0, 1, c->stream, c->d_row_list, (int)again.size());
                } else {
                    Timed t(c, F_RANK_TIED, (2.0 + 2.0) * (double)n * (double)n * part);
                    launch_rank_rows_tied(c->dR, c->d_tie_bits, c->ld_bits, c->d_order, c->d_order + n, (int)n, c->d_row_list,
                                          (int)again.size(), c->dRank, ldr, c->stream, c->d_row_list + n);
                }
            }
            HIPCHK(hipGetLastError());
            tick();
            HIPCHK(sync_stream(c));
            tick();
            report();
            c->have_rank = true;
            c->cached_start = -1;
            return HICMI_OK;
        }
    }
    c->presort_n = 0;                                               // (dR is overwritten below)
    if (!bitonic) {
        Timed t(c, F_SORT, (8.0 + 2.0) * (double)n * (double)n * share);
        launch_rank_rows_radix(c->dC, c->ldc, c->d_order, c->d_order + n, c->d_np, c->d_seq, (int)n, c->d_sort_scratch, c->dRank,
                               ldr, (int)c->shard_first, (int)c->shard_stride, c->stream);
    } else {
        {
            Timed t(c, F_SORT, (8.0 + 2.0) * (double)n * (double)n * share);
            launch_sort_rows(c->dC, c->ldc, c->d_order, c->d_order + n, c->d_np, c->d_seq, (int)n, c->d_sort_scratch, c->dR, ldr,
                             (int)c->shard_first, (int)c->shard_stride, c->stream);
        }
        HIPCHK(hipGetLastError());
        Timed t(c, F_RANK_INVERT, (2.0 + 2.0) * (double)n * (double)n * share);
        launch_rank_invert(c->dR, c->dRank, ldr, (int)n, (int)c->shard_first, (int)c->shard_stride, c->stream);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(sync_stream(c));
    c->have_rank = true;
    c->cached_start = -1;
    return HICMI_OK;
}

int hicmi_presort_state(hicmi_ctx* c, int* state_out, int64_t* tied_rows_out)
{
    if (!c || !state_out) return fail(HICMI_EINVAL, "bad arguments");
    *state_out = c->presort_used;
    if (tied_rows_out) *tied_rows_out = c->presort_used ? c->presort_tied_rows : 0;
    return HICMI_OK;
}

int hicmi_get_rank_rows(hicmi_ctx* c, int64_t row0, int64_t nrows, int inverse, uint16_t* out)
{
    if (!c || !out || row0 < 0 || nrows < 0) return fail(HICMI_EINVAL, "bad arguments");
    if (!c->have_rank) return fail(HICMI_EINVAL, "hicmi_rank_matrix has not run");
    if (row0 + nrows > c->n) return fail(HICMI_EINVAL, "rows out of range");
    HIPCHK(hipSetDevice(c->device));
    // the device keeps the RANK rows (position of every column); an argsort row is their inverse, formed here on request
    const uint16_t* src = c->dRank + row0 * c->ldr;
    HIPCHK(hipMemcpy2DAsync(out, sizeof(uint16_t) * (size_t)c->n, src, sizeof(uint16_t) * (size_t)c->ldr,
                            sizeof(uint16_t) * (size_t)c->n, (size_t)nrows, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(sync_stream(c));
    if (!inverse) {
        std::vector<uint16_t> tmp((size_t)c->n);
        for (int64_t r = 0; r < nrows; r++) {
            uint16_t* row = out + r * c->n;
            for (int64_t col = 0; col < c->n; col++) {
                if (row[col] >= c->n) return fail(HICMI_ESTATE, "rank row %lld is not a permutation (a row of another shard?)", (long long)(row0 + r));
                tmp[row[col]] = (uint16_t)col;
            }
            memcpy(row, tmp.data(), sizeof(uint16_t) * (size_t)c->n);
        }
    }
    return HICMI_OK;
}

int hicmi_get_similarity_row(hicmi_ctx* c, int64_t row, double* out)
{
    if (!c || !out) return fail(HICMI_EINVAL, "bad arguments");
    if (!c->have_rank) return fail(HICMI_EINVAL, "hicmi_rank_matrix has not run");
    if (row < 0 || row >= c->n) return fail(HICMI_EINVAL, "row out of range");
    HIPCHK(hipSetDevice(c->device));
    int rc = ensure(c->d_tmp, c->tmp_cap, c->n);
    if (rc) return rc;
    launch_similarity_row(c->dC, c->ldc, c->d_order, c->d_np, c->d_seq, (int)c->n, (int)row, c->d_tmp, c->stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, c->d_tmp, sizeof(double) * (size_t)c->n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(sync_stream(c));
    return HICMI_OK;
}

// ---------------------------------------------------------------------------------------------------
static int ensure_scan_buffers(hicmi_ctx* c)
{
    if (c->x_cap >= c->n && c->d_x && c->d_sig) return HICMI_OK;
    free_dev(c->d_x); free_dev(c->d_sig); c->d_x = nullptr; c->d_sig = nullptr; c->x_cap = 0;
    HIPCHK(hipMalloc((void**)&c->d_x, sizeof(int32_t) * (size_t)c->n));
    HIPCHK(hipMalloc((void**)&c->d_sig, (size_t)c->n));
    c->x_cap = c->n;
    return HICMI_OK;
}

// The significance flags of the rows counted in d_x.  The flags are all the host loops need, tens of thousands of
// times per map: the kernel writes them straight into the pinned buffer (no copy kernel, one synchronisation).
static int finish_scan(hicmi_ctx* c, int64_t rows, int mode, int64_t L_fixed, int64_t M, double psig, int32_t* x_out,
                       uint8_t* sig_out, int64_t own_first)
{
    const bool direct = sig_out && !x_out;
    if (direct) { int rc = ensure_pin_down(c, (size_t)rows); if (rc) return rc; }
    {
        Timed t(c, F_HYPER_FLAGS, 5.0 * (double)rows);
        launch_hyper_flags(c->d_x, (int)rows, mode, (int)L_fixed, M, psig, direct ? reinterpret_cast<uint8_t*>(c->pin_down) : c->d_sig,
                           (int)own_first, (int)c->shard_stride, c->stream);
    }
    HIPCHK(hipGetLastError());
    if (direct) {
        HIPCHK(sync_stream(c));
        memcpy(sig_out, c->pin_down, (size_t)rows);
        return HICMI_OK;
    }
    if (x_out) { int rc = download(c, x_out, c->d_x, sizeof(int32_t) * (size_t)rows); if (rc) return rc; }
    if (sig_out) { int rc = download(c, sig_out, c->d_sig, (size_t)rows); if (rc) return rc; }
    if (!x_out && !sig_out) HIPCHK(sync_stream(c));
    return HICMI_OK;
}

int hicmi_cut_scan(hicmi_ctx* c, int64_t start, int64_t M, double psig, int32_t* x_out, uint8_t* sig_out)
{
    if (!c) return fail(HICMI_EINVAL, "NULL context");
    if (!c->have_rank) return fail(HICMI_EINVAL, "hicmi_rank_matrix has not run");
    const int64_t n = c->n;
    if (start < 0 || start >= n) return fail(HICMI_EINVAL, "start out of range");
    HIPCHK(hipSetDevice(c->device));
    int rc = ensure_scan_buffers(c);
    if (rc) return rc;
    const int64_t cnt = n - start;              // entries: rows start .. n-1
    const int64_t st = c->shard_stride;
    const int64_t t0 = ((c->shard_first - start) % st + st) % st;      // first entry whose row this shard owns
    if (c->cached_start != start) {
        HIPCHK(hipMemsetAsync(c->d_x, 0, sizeof(int32_t) * (size_t)(st > 1 ? cnt : 1), c->stream));
        const double m = (double)(cnt - 1);
        Timed t(c, F_CUT_COUNT, 2.0 * (m * (m + 3.0) / 2.0) / (double)st);      // sum_{L=1..m} (L+1) uint16 entries
        const int64_t t1 = t0 >= 1 ? t0 : t0 + st;                      // entry 0 (the row `start` itself) is never counted
        if (t1 < cnt)
            launch_cut_count(c->dRank, c->ldr, (int)(start + t1), (int)(cnt - t1), (int)start, 0, 0, c->d_x + t1, (int)st, c->stream);
        HIPCHK(hipGetLastError());
        c->cached_start = start;
    }
    return finish_scan(c, cnt, 0, 0, M, psig, x_out, sig_out, t0);
}

int hicmi_filter_scan(hicmi_ctx* c, int64_t start, int64_t cut, int64_t n_rows, int64_t M, double psig,
                      int32_t* x_out, uint8_t* sig_out)
{
    if (!c) return fail(HICMI_EINVAL, "NULL context");
    if (!c->have_rank) return fail(HICMI_EINVAL, "hicmi_rank_matrix has not run");
    const int64_t n = c->n;
    if (start < 0 || start >= n || cut < start || cut >= n || n_rows < 0 || start + n_rows > n)
        return fail(HICMI_EINVAL, "filter scan arguments out of range");
    HIPCHK(hipSetDevice(c->device));
    int rc = ensure_scan_buffers(c);
    if (rc) return rc;
    c->cached_start = -1;                       // d_x is reused
    const int64_t st = c->shard_stride;
    const int64_t t0 = ((c->shard_first - start) % st + st) % st;      // first entry whose row this shard owns
    if (st > 1) HIPCHK(hipMemsetAsync(c->d_x, 0, sizeof(int32_t) * (size_t)n_rows, c->stream));
    {
        Timed t(c, F_CUT_COUNT, 2.0 * (double)n_rows * (double)(cut - start + 1) / (double)st);
        if (t0 < n_rows)
            launch_cut_count(c->dRank, c->ldr, (int)(start + t0), (int)(n_rows - t0), (int)start, 1, (int)cut, c->d_x + t0, (int)st,
                             c->stream);
    }
    HIPCHK(hipGetLastError());
    return finish_scan(c, n_rows, 1, cut - start, M, psig, x_out, sig_out, t0);
}

double hicmi_hypergeom_sf(int64_t x, int64_t M, int64_t n, int64_t N) { return hypergeom_sf_ge(x, M, n, N); }
int hicmi_hypergeom_decide(int64_t x, int64_t M, int64_t n, int64_t N, double psig) { return hypergeom_decide(x, M, n, N, psig); }

// ---- the two scan loops with their control flow on the device (k_part1_scan.hip) -------------------------------------
// One allocation: [ScanState, 256 B][cuts n][M log 2n][alt n][seg 3n][seg_x n] int32, then [filt n][prev n] bytes.
static int ensure_scan_program(hicmi_ctx* c)
{
    if (c->scan_prog_cap >= c->n && c->d_scan_prog) return HICMI_OK;
    free_dev(c->d_scan_prog); c->d_scan_prog = nullptr; c->scan_prog_cap = 0;
    HIPCHK(hipMalloc((void**)&c->d_scan_prog, 256 + sizeof(int32_t) * 8 * (size_t)c->n + 2 * (size_t)c->n + 64));
    c->scan_prog_cap = c->n;
    return HICMI_OK;
}

struct ScanProgram {
    ScanState* st; int32_t *cuts, *mlog, *alt, *seg, *seg_x; uint8_t *filt, *prev;
    explicit ScanProgram(hicmi_ctx* c)
    {
        unsigned char* b = c->d_scan_prog;
        const size_t n = (size_t)c->scan_prog_cap;
        st = reinterpret_cast<ScanState*>(b);
        cuts = reinterpret_cast<int32_t*>(b + 256);
        mlog = cuts + n; alt = mlog + 2 * n; seg = alt + n; seg_x = seg + 3 * n;
        filt = reinterpret_cast<uint8_t*>(seg_x + n); prev = filt + n;
    }
};

// Batches of scans until the device says the loop has ended.  `pairs` launches per batch: the record is read once per batch.
static int run_scan_program(hicmi_ctx* c, ScanState& h, const ScanProgram& p, int64_t max_scans,
                            const std::function<void(int)>& enqueue)
{
    int rc = upload(c, p.st, &h, sizeof(h));
    if (rc) return rc;
    const int pairs = 32;
    int64_t batches = 0;
    for (int64_t issued = 0; ; issued += pairs) {
        if (issued > max_scans + pairs) return fail(HICMI_ESTATE, "scan loop did not end after %lld scans", (long long)issued);
        {
            Timed t(c, F_CUT_COUNT, 0.0);
            enqueue(pairs);
        }
        HIPCHK(hipGetLastError());
        batches++;
        rc = download(c, &h, p.st, sizeof(h));
        if (rc) return rc;
        if (h.done) break;
    }
    c->launches[F_CUT_COUNT] += (int64_t)h.scans - batches;       // the family is reported per scan
    c->launches[F_HYPER_FLAGS] += (int64_t)h.scans;               // (the decisions ride in the same launches)
    c->bytes[F_CUT_COUNT] += (double)h.bytes;
    c->cached_start = -1;                                          // d_x was reused
    return HICMI_OK;
}

int hicmi_first_pass_cuts(hicmi_ctx* c, int64_t min_size, int64_t stop_ind, double psig, int32_t* cuts_out, int64_t cuts_cap,
                          int64_t* n_cuts_out, int32_t* m_log_out, int64_t log_cap, int64_t* n_log_out)
{
    if (!c || !cuts_out || !n_cuts_out || !n_log_out || (log_cap > 0 && !m_log_out)) return fail(HICMI_EINVAL, "bad arguments");
    if (!c->have_rank) return fail(HICMI_EINVAL, "hicmi_rank_matrix has not run");
    if (c->shard_stride != 1) return fail(HICMI_EINVAL, "the device-driven scan loops need the whole rank matrix (no row shard)");
    if (min_size < 1) return fail(HICMI_EINVAL, "min_size must be >= 1");
    const int64_t n = c->n;
    HIPCHK(hipSetDevice(c->device));
    int rc = ensure_scan_buffers(c);
    if (rc) return rc;
    rc = ensure_scan_program(c);
    if (rc) return rc;
    ScanProgram p(c);
    ScanState h;
    memset(&h, 0, sizeof(h));
    h.mode = 0; h.start = 0; h.M = n; h.recount = 1;
    h.min_size = (int)std::min<int64_t>(min_size, n + 1); h.stop_ind = (int)std::min<int64_t>(stop_ind, INT32_MAX);
    const int lcap = (int)n;                                       // pairs that fit the device log
    rc = run_scan_program(c, h, p, 6 * n, [&](int pairs) {
        launch_first_pass_pairs(c->dRank, c->ldr, (int)n, p.st, c->d_x, c->d_sig, psig, p.cuts, p.mlog, lcap, pairs, c->stream);
    });
    if (rc) return rc;
    if (h.n_cuts > cuts_cap) return fail(HICMI_EINVAL, "%d cuts do not fit cuts_cap", h.n_cuts);
    if (h.n_log > lcap || h.n_log > log_cap) return fail(HICMI_EINVAL, "%d M changes do not fit the log", h.n_log);
    if (h.n_cuts) { rc = download(c, cuts_out, p.cuts, sizeof(int32_t) * (size_t)h.n_cuts); if (rc) return rc; }
    if (h.n_log) { rc = download(c, m_log_out, p.mlog, sizeof(int32_t) * 2 * (size_t)h.n_log); if (rc) return rc; }
    *n_cuts_out = h.n_cuts; *n_log_out = h.n_log;
    return HICMI_OK;
}

int hicmi_filter_cuts(hicmi_ctx* c, const int32_t* cuts_in, int64_t n_in, double psig, int32_t* cuts_out, int64_t cuts_cap,
                      int64_t* n_out, int64_t* warned_out)
{
    if (!c || !cuts_in || !cuts_out || !n_out || n_in < 1) return fail(HICMI_EINVAL, "bad arguments");
    if (!c->have_rank) return fail(HICMI_EINVAL, "hicmi_rank_matrix has not run");
    if (c->shard_stride != 1) return fail(HICMI_EINVAL, "the device-driven scan loops need the whole rank matrix (no row shard)");
    const int64_t n = c->n;
    if (n_in > n) return fail(HICMI_EINVAL, "more cuts than rows");
    for (int64_t i = 0; i < n_in; i++)
        if (cuts_in[i] < 0 || cuts_in[i] >= n || (i && cuts_in[i] <= cuts_in[i - 1]))
            return fail(HICMI_EINVAL, "cuts must be ascending indices in [0, n)");
    HIPCHK(hipSetDevice(c->device));
    int rc = ensure_scan_buffers(c);
    if (rc) return rc;
    rc = ensure_scan_program(c);
    if (rc) return rc;
    ScanProgram p(c);
    rc = upload(c, p.alt, cuts_in, sizeof(int32_t) * (size_t)n_in);
    if (rc) return rc;
    HIPCHK(hipMemsetAsync(p.filt, 0, 2 * (size_t)c->scan_prog_cap, c->stream));      // filtered = {}, prev_filtered = {}
    ScanState h;
    memset(&h, 0, sizeof(h));
    h.mode = 1; h.start = 0; h.M = n; h.recount = 1;
    h.MD = (int)(n / 5);                                           // MD = int(n / 5)  (S2C:575)
    h.n_alt = (int)n_in; h.f_max_rounds = (int)std::min<int64_t>(10 * n_in, INT32_MAX);   // S2C:577
    h.cut = cuts_in[0];
    h.n_rows = (int)std::min<int64_t>(n, (int64_t)h.MD + 1);
    const int max_rows = (int)std::min<int64_t>(n, (int64_t)h.MD + 1);
    // every pass over the candidates runs at most 10 * n_in rounds of at most n_in scans; the passes end when the set
    // of kept cuts repeats - bounded here far above anything a map produces
    const int64_t max_scans = std::min<int64_t>((int64_t)4000000, 20 * n_in * n_in * 10 + 1000);
    rc = run_scan_program(c, h, p, max_scans, [&](int pairs) {
        launch_filter_pairs(c->dRank, c->ldr, (int)n, max_rows, p.st, c->d_x, c->d_sig, psig, p.alt, p.filt, p.prev, p.seg,
                            p.seg_x, pairs, c->stream);
    });
    if (rc) return rc;
    std::vector<uint8_t> kept((size_t)n);
    rc = download(c, kept.data(), p.filt, (size_t)n);
    if (rc) return rc;
    int64_t m = 0;
    for (int64_t e = 0; e < n; e++)
        if (kept[(size_t)e]) { if (m >= cuts_cap) return fail(HICMI_EINVAL, "filtered cuts do not fit cuts_cap"); cuts_out[m++] = (int32_t)e; }
    *n_out = m;
    if (warned_out) *warned_out = h.f_warned;
    return HICMI_OK;
}

// ---------------------------------------------------------------------------------------------------
int hicmi_p2_select(hicmi_ctx* c, const int32_t* sel, int64_t n)
{
    if (!c || !sel || n < 1) return fail(HICMI_EINVAL, "bad arguments");
    if (!c->dC) return fail(HICMI_EINVAL, "no contact matrix set");
    for (int64_t i = 0; i < n; i++)
        if (sel[i] < 0 || sel[i] >= c->n) return fail(HICMI_EINVAL, "selection index out of range");
    HIPCHK(hipSetDevice(c->device));
    const int64_t ld2 = (n + 15) & ~(int64_t)15;
    int rc = ensure(c->dM2, c->m2_cap, n * ld2);
    if (rc) return rc;
    rc = ensure(c->d_sel, c->sel_cap, n);
    if (rc) return rc;
    // H[k] = 1 + 1/2 + ... + 1/k, left to right in fp64
    if (c->h_cap < n + 1) {
        std::vector<double> H((size_t)(n + 1));
        H[0] = 0.0;
        for (int64_t k = 1; k <= n; k++) H[(size_t)k] = H[(size_t)k - 1] + 1.0 / (double)k;
        rc = ensure(c->d_H, c->h_cap, n + 1);
        if (rc) return rc;
        rc = upload(c, c->d_H, H.data(), sizeof(double) * (size_t)(n + 1));
        if (rc) return rc;
        HIPCHK(sync_stream(c));
    }
    rc = upload(c, c->d_sel, sel, sizeof(int32_t) * (size_t)n);
    if (rc) return rc;
    {
        Timed t(c, F_P2_SELECT, 16.0 * (double)n * (double)n);
        launch_p2_select(c->dC, c->ldc, c->d_sel, (int)n, c->dM2, ld2, c->stream);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(sync_stream(c));
    c->n2 = n; c->ld2 = ld2;
    c->n_scaf = 0; c->n_arr = 0; c->h_arr_id.clear();
    c->cache_valid = false; c->exact_cache.clear();
    return HICMI_OK;
}

int hicmi_p2_total(hicmi_ctx* c, double* total_out)
{
    if (!c || !total_out) return fail(HICMI_EINVAL, "bad arguments");
    if (c->n2 < 1) return fail(HICMI_EINVAL, "hicmi_p2_select has not run");
    HIPCHK(hipSetDevice(c->device));
    int rc = ensure(c->d_partial, c->partial_cap, c->n2 + 1);
    if (rc) return rc;
    {
        Timed t(c, F_P2_TOTAL, 4.0 * (double)c->n2 * (double)c->n2);
        launch_p2_total(c->dM2, c->ld2, (int)c->n2, c->d_partial, c->d_partial + c->n2, c->stream);
    }
    HIPCHK(hipGetLastError());
    return download(c, total_out, c->d_partial + c->n2, sizeof(double));
}

int hicmi_p2_score(hicmi_ctx* c, const int32_t* perms, int64_t n_cand, int64_t n_used, double total, double* scores_out)
{
    if (!c || !perms || !scores_out || n_cand < 0 || n_used < 1) return fail(HICMI_EINVAL, "bad arguments");
    if (c->n2 < 1) return fail(HICMI_EINVAL, "hicmi_p2_select has not run");
    if (n_used > c->n2) return fail(HICMI_EINVAL, "n_used exceeds the selection");
    if (n_cand == 0) return HICMI_OK;
    if (n_used * (int64_t)sizeof(int32_t) > 160 * 1024) return fail(HICMI_EUNSUPPORTED, "candidate longer than 40960 bins");
    for (int64_t i = 0; i < n_cand * n_used; i++)
        if (perms[i] < 0 || perms[i] >= c->n2) return fail(HICMI_EINVAL, "candidate index out of range");
    HIPCHK(hipSetDevice(c->device));
    int rc = ensure(c->d_perms, c->perms_cap, n_cand * n_used);
    if (rc) return rc;
    rc = ensure(c->d_scores, c->scores_cap, n_cand);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(c->d_perms, perms, sizeof(int32_t) * (size_t)(n_cand * n_used), hipMemcpyHostToDevice, c->stream));
    {
        Timed t(c, F_P2_SCORE, 8.0 * (double)n_cand * 0.5 * (double)n_used * (double)(n_used - 1));
        launch_p2_score(c->dM2, c->ld2, c->d_perms, (int)n_cand, (int)n_used, c->d_H, 0.0, total, c->d_scores, c->stream);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(scores_out, c->d_scores, sizeof(double) * (size_t)n_cand, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(sync_stream(c));
    return HICMI_OK;
}

int hicmi_p2_score_exact(hicmi_ctx* c, const int32_t* perms, int64_t n_cand, int64_t n_used, double total,
                         double* scores_out)
{
    if (!c || !perms || !scores_out || n_cand < 0 || n_used < 1) return fail(HICMI_EINVAL, "bad arguments");
    if (c->n2 < 1) return fail(HICMI_EINVAL, "hicmi_p2_select has not run");
    if (n_used > c->n2) return fail(HICMI_EINVAL, "n_used exceeds the selection");
    if (n_cand == 0) return HICMI_OK;
    if (n_used * (int64_t)sizeof(int32_t) > 160 * 1024) return fail(HICMI_EUNSUPPORTED, "candidate longer than 40960 bins");
    for (int64_t i = 0; i < n_cand * n_used; i++)
        if (perms[i] < 0 || perms[i] >= c->n2) return fail(HICMI_EINVAL, "candidate index out of range");
    HIPCHK(hipSetDevice(c->device));
    int rc = ensure(c->d_perms, c->perms_cap, n_cand * n_used);
    if (rc) return rc;
    rc = ensure(c->d_scores, c->scores_cap, n_cand);
    if (rc) return rc;
    rc = ensure(c->d_T, c->t_cap, 2 * n_cand * n_used);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(c->d_perms, perms, sizeof(int32_t) * (size_t)(n_cand * n_used), hipMemcpyHostToDevice, c->stream));
    {
        Timed t(c, F_P2_EXACT, 8.0 * (double)n_cand * 0.5 * (double)n_used * (double)(n_used - 1));
        launch_p2_score_exact(c->dM2, c->ld2, c->d_perms, (int)n_cand, (int)n_used, total, c->d_T,
                              c->d_T + n_cand * n_used, c->d_scores, c->stream);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(scores_out, c->d_scores, sizeof(double) * (size_t)n_cand, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(sync_stream(c));
    return HICMI_OK;
}


// ---------------------------------------------------------------------------------------------------
// Part 2 search with device-side candidate enumeration (k_part2_search.hip)
int hicmi_p2_layout(hicmi_ctx* c, const int32_t* scaf_start, const int32_t* scaf_len, int64_t n_scaf)
{
    if (!c || !scaf_start || !scaf_len || n_scaf < 1) return fail(HICMI_EINVAL, "bad arguments");
    if (c->n2 < 1) return fail(HICMI_EINVAL, "hicmi_p2_select has not run");
    for (int64_t i = 0; i < n_scaf; i++)
        if (scaf_len[i] < 1 || scaf_start[i] < 0 || (int64_t)scaf_start[i] + scaf_len[i] > c->n2)
            return fail(HICMI_EINVAL, "scaffold %lld is not a range of the selection", (long long)i);
    HIPCHK(hipSetDevice(c->device));
    if (c->scaf_cap < n_scaf) {
        free_dev(c->d_scaf_start); free_dev(c->d_scaf_len); c->d_scaf_start = c->d_scaf_len = nullptr; c->scaf_cap = 0;
        HIPCHK(hipMalloc((void**)&c->d_scaf_start, sizeof(int32_t) * (size_t)n_scaf));
        HIPCHK(hipMalloc((void**)&c->d_scaf_len, sizeof(int32_t) * (size_t)n_scaf));
        c->scaf_cap = n_scaf;
    }
    c->h_scaf_start.assign(scaf_start, scaf_start + n_scaf);
    c->h_scaf_len.assign(scaf_len, scaf_len + n_scaf);
    c->n_scaf = n_scaf; c->n_arr = 0;
    int rc_up = upload(c, c->d_scaf_start, scaf_start, sizeof(int32_t) * (size_t)n_scaf);
    if (rc_up) return rc_up;
    return upload(c, c->d_scaf_len, scaf_len, sizeof(int32_t) * (size_t)n_scaf);     // consumed in stream order
}

int hicmi_p2_set_arrangement(hicmi_ctx* c, const int32_t* ids, const uint8_t* rev, int64_t S)
{
    if (!c || !ids || !rev || S < 1) return fail(HICMI_EINVAL, "bad arguments");
    if (c->n_scaf < 1) return fail(HICMI_EINVAL, "hicmi_p2_layout has not run");
    std::vector<int32_t> pos((size_t)S + 1, 0);
    std::vector<uint8_t> used((size_t)c->n_scaf, 0);
    for (int64_t j = 0; j < S; j++) {
        if (ids[j] < 0 || ids[j] >= c->n_scaf || used[(size_t)ids[j]]) return fail(HICMI_EINVAL, "arrangement must list distinct scaffolds of the layout");
        used[(size_t)ids[j]] = 1;
        pos[(size_t)j + 1] = pos[(size_t)j] + c->h_scaf_len[(size_t)ids[j]];
    }
    HIPCHK(hipSetDevice(c->device));
    int rc = ensure(c->d_arr_packed, c->arr_cap, 3 * std::max<int64_t>(S, c->n_scaf) + 2);
    if (rc) return rc;
    rc = ensure(c->d_pos2sel, c->pos_cap, c->n2);
    if (rc) return rc;
    c->h_arr_id.assign(ids, ids + S); c->h_arr_rev.assign(rev, rev + S); c->h_arr_pos = pos;
    c->arr_version++;
    c->n_arr = pos[(size_t)S];
    c->h_pos2sel.resize((size_t)c->n_arr);
    for (int64_t j = 0; j < S; j++) {
        const int32_t st = c->h_scaf_start[(size_t)ids[j]], ln = c->h_scaf_len[(size_t)ids[j]];
        int32_t* dst = c->h_pos2sel.data() + pos[(size_t)j];
        if (rev[j]) for (int32_t e = 0; e < ln; e++) dst[e] = st + ln - 1 - e;
        else        for (int32_t e = 0; e < ln; e++) dst[e] = st + e;
    }
    c->h_arr_packed.resize((size_t)(3 * S + 1));
    for (int64_t j = 0; j < S; j++) { c->h_arr_packed[(size_t)j] = ids[j]; c->h_arr_packed[(size_t)(2 * S + 1 + j)] = rev[j] ? 1 : 0; }
    for (int64_t j = 0; j <= S; j++) c->h_arr_packed[(size_t)(S + j)] = pos[(size_t)j];
    rc = upload(c, c->d_arr_packed, c->h_arr_packed.data(), sizeof(int32_t) * (size_t)(3 * S + 1));
    if (rc) return rc;
    launch_arr_materialize(c->d_arr_packed, (int)S, c->d_scaf_start, c->d_scaf_len, (int)c->n_arr, c->d_pos2sel, c->stream);
    HIPCHK(hipGetLastError());
    return HICMI_OK;
}

int hicmi_p2_arrangement_total(hicmi_ctx* c, double* total_out)
{
    if (!c || !total_out) return fail(HICMI_EINVAL, "bad arguments");
    if (c->n_arr < 1) return fail(HICMI_EINVAL, "hicmi_p2_set_arrangement has not run");
    HIPCHK(hipSetDevice(c->device));
    int rc = ensure(c->d_T, c->t_cap, c->n_arr + 1);
    if (rc) return rc;
    rc = ensure(c->d_scores, c->scores_cap, 1);
    if (rc) return rc;
    {
        Timed t(c, F_P2_TOTAL, 4.0 * (double)c->n_arr * (double)c->n_arr);
        launch_p2_total_perm(c->dM2, c->ld2, c->d_pos2sel, (int)c->n_arr, c->d_T, c->d_scores, c->stream);
    }
    HIPCHK(hipGetLastError());
    rc = download(c, total_out, c->d_scores, sizeof(double));
    if (rc) return rc;
    return HICMI_OK;
}

int hicmi_p2_arrangement_score(hicmi_ctx* c, double total, double* score_out)
{
    if (!c || !score_out) return fail(HICMI_EINVAL, "bad arguments");
    if (c->n_arr < 1) return fail(HICMI_EINVAL, "hicmi_p2_set_arrangement has not run");
    if (c->n_arr < 2) { *score_out = 0.0; return HICMI_OK; }
    HIPCHK(hipSetDevice(c->device));
    const int NB = kBaseSlabs;
    int rc = ensure(c->d_scores, c->scores_cap, NB);
    if (rc) return rc;
    {
        Timed t(c, F_P2_SCORE, 4.0 * (double)c->n_arr * (double)(c->n_arr - 1));
        launch_p2_base_partial(c->dM2, c->ld2, c->d_pos2sel, (int)c->n_arr, c->d_H, (int)c->n_arr, NB, c->d_scores, c->stream);
    }
    HIPCHK(hipGetLastError());
    double part[NB];
    rc = download(c, part, c->d_scores, sizeof(double) * NB);
    if (rc) return rc;
    double sum = 0.0;
    for (int b = 0; b < NB; b++) sum += part[b];
    *score_out = sum / total;
    return HICMI_OK;
}

int hicmi_p2_score_insertions(hicmi_ctx* c, int32_t new_id, double total, double* scores_out)
{
    if (!c || !scores_out) return fail(HICMI_EINVAL, "bad arguments");
    if (c->n_scaf < 1) return fail(HICMI_EINVAL, "hicmi_p2_layout has not run");
    if (new_id < 0 || new_id >= c->n_scaf) return fail(HICMI_EINVAL, "new scaffold out of range");
    const int64_t S = (int64_t)c->h_arr_id.size();
    if (c->n_arr < 1 || S < 1) return fail(HICMI_EINVAL, "hicmi_p2_set_arrangement has not run");
    for (int64_t j = 0; j < S; j++) if (c->h_arr_id[(size_t)j] == new_id) return fail(HICMI_EINVAL, "scaffold is already in the arrangement");
    const int new_len = c->h_scaf_len[(size_t)new_id];
    if ((c->n_arr + new_len) * (int64_t)sizeof(int32_t) > 160 * 1024) return fail(HICMI_EUNSUPPORTED, "candidate longer than 40960 bins");
    HIPCHK(hipSetDevice(c->device));
    // incremental form: BASE (64 partial sums) - STRADDLE(g) (prefix sums of S increments) + CROSS(g, r)
    const int NB = kBaseSlabs;
    const int64_t n_out = NB + S + 2 * (S + 1);
    int rc = ensure(c->d_scores, c->scores_cap, n_out);
    if (rc) return rc;
    {
        const double nn = (double)c->n_arr;
        Timed t(c, F_P2_INSERT, 8.0 * (0.5 * nn * (nn - 1.0) + nn * nn + 2.0 * (double)(S + 1) * (double)new_len * nn));
        launch_p2_insert_delta(c->dM2, c->ld2, c->d_pos2sel, (int)c->n_arr, c->d_arr_packed + S, (int)S,
                               c->h_scaf_start[(size_t)new_id], new_len, c->d_H, NB, c->d_scores, c->stream);
    }
    HIPCHK(hipGetLastError());
    std::vector<double> host((size_t)n_out);
    rc = download(c, host.data(), c->d_scores, sizeof(double) * (size_t)n_out);
    if (rc) return rc;
    double base = 0.0;
    for (int b = 0; b < NB; b++) base += host[(size_t)b];
    double straddle = 0.0;
    for (int64_t g = 0; g <= S; g++) {
        if (g > 0) straddle += host[(size_t)(NB + g - 1)];
        for (int r = 0; r < 2; r++)
            scores_out[2 * g + r] = (base - straddle + host[(size_t)(NB + S + 2 * g + r)]) / total;
    }
    return HICMI_OK;
}

int hicmi_p2_window_tables(hicmi_ctx* c, int64_t k, const int8_t* orders, int64_t n_orders, const uint8_t* orients,
                           int64_t n_orients)
{
    if (!c || !orders || !orients || k < 1 || k > 8 || n_orders < 1 || n_orients < 1) return fail(HICMI_EINVAL, "bad arguments");
    for (int64_t i = 0; i < n_orders * k; i++) if (orders[i] < 0 || orders[i] >= k) return fail(HICMI_EINVAL, "order table entry out of range");
    HIPCHK(hipSetDevice(c->device));
    int rc = ensure(c->d_orders, c->ord_cap, n_orders * k);
    if (rc) return rc;
    rc = ensure(c->d_orients, c->ori_cap, n_orients * k);
    if (rc) return rc;
    rc = upload(c, c->d_orders, orders, (size_t)(n_orders * k));
    if (rc) return rc;
    rc = upload(c, c->d_orients, orients, (size_t)(n_orients * k));
    if (rc) return rc;
    c->tab_k = (int)k; c->n_orders = n_orders; c->n_orients = n_orients;
    c->h_orders.assign(orders, orders + n_orders * k);
    c->h_orients.assign(orients, orients + n_orients * k);
    return HICMI_OK;
}

namespace {
// deltas of `count` consecutive windows (first0, first0+1, ...) of k scaffolds against the CURRENT
// arrangement, one launch pair; delta_out: count x n_cand
int window_batch(hicmi_ctx* c, int64_t first0, int64_t count, int64_t k, double* delta_out)
{
    const int64_t S = (int64_t)c->h_arr_id.size();
    if (c->n_arr < 1 || S < 1) return fail(HICMI_EINVAL, "hicmi_p2_set_arrangement has not run");
    if (k != c->tab_k) return fail(HICMI_EINVAL, "hicmi_p2_window_tables has not been called for k = %lld", (long long)k);
    if (first0 < 0 || count < 1 || first0 + count - 1 + k > S) return fail(HICMI_EINVAL, "window out of range");
    HIPCHK(hipSetDevice(c->device));
    const int64_t n_cand = c->n_orders * c->n_orients;
    std::vector<WindowBatchEntry> wb((size_t)count);
    int64_t g_total = 0; int max_m = 0; double g_bytes = 0.0, d_bytes = 0.0, g_flops = 0.0;
    for (int64_t wdx = 0; wdx < count; wdx++) {
        const int64_t first = first0 + wdx;
        WindowBatchEntry& e = wb[(size_t)wdx];
        memset(&e, 0, sizeof(e));
        const int p0 = c->h_arr_pos[(size_t)first], p1 = c->h_arr_pos[(size_t)(first + k)];
        e.p0 = p0; e.m = p1 - p0; e.g_off = g_total;
        for (int64_t j = 0; j < k; j++) {
            const int32_t sc = c->h_arr_id[(size_t)(first + j)];
            e.w.start[j] = c->h_scaf_start[(size_t)sc];
            e.w.len[j] = c->h_scaf_len[(size_t)sc];
            e.w.off[j] = c->h_arr_pos[(size_t)(first + j)] - p0;
            e.w.rev[j] = c->h_arr_rev[(size_t)(first + j)];
        }
        g_total += (int64_t)e.m * e.m;
        max_m = std::max(max_m, e.m);
        g_bytes += 8.0 * (double)e.m * (double)(c->n_arr - e.m);
        g_flops += 2.0 * (double)e.m * (double)e.m * (double)(c->n_arr - e.m);       // A (m x (n - m)) . Toeplitz ((n - m) x m), all scaffolds of the window
        d_bytes += 8.0 * (double)n_cand * (0.5 * (double)e.m * (double)(e.m - 1) + (double)e.m);
    }
    // placement tables (k_part2_window.hip) unless the direct per-candidate kernels are asked for (A/B switch)
    static const bool direct = getenv("HICMI_P2_WINDOW_DIRECT") != nullptr;
    if (!direct) g_total = count * window_table_doubles((int)k);
    int rc = ensure(c->d_G, c->g_cap, g_total);
    if (rc) return rc;
    rc = ensure(c->d_delta, c->delta_cap, n_cand * count);
    if (rc) return rc;
    rc = ensure(c->d_wb, c->wb_cap, count);
    if (rc) return rc;
    rc = upload(c, c->d_wb, wb.data(), sizeof(WindowBatchEntry) * (size_t)count);
    if (rc) return rc;
    {
        // the G and delta kernels are launched as a pair; their algorithmic bytes are booked separately
        c->launches[F_P2_WINDOW_DELTA]++; c->bytes[F_P2_WINDOW_DELTA] += d_bytes;
        if (!direct) { c->launches[F_P2_WINDOW_FLOPS]++; c->bytes[F_P2_WINDOW_FLOPS] += g_flops; }   // (flops of the GEMM form: bench.py's `mfma`)
        Timed t(c, F_P2_WINDOW_G, g_bytes);
        if (direct)
            launch_p2_window_batch(c->dM2, c->ld2, c->d_pos2sel, (int)c->n_arr, (int)k, c->d_wb, (int)count, max_m, c->d_orders,
                                   c->d_orients, (int)c->n_orders, (int)c->n_orients, c->d_H, c->d_G, c->d_delta, c->stream);
        else
            launch_p2_window_tables(c->dM2, c->ld2, c->d_pos2sel, (int)c->n_arr, (int)k, c->d_wb, wb.data(), (int)count, max_m,
                                    c->d_orders, c->d_orients, (int)c->n_orders, (int)c->n_orients, c->d_H, c->d_G, c->d_delta,
                                    c->stream);
    }
    HIPCHK(hipGetLastError());
    rc = download(c, delta_out, c->d_delta, sizeof(double) * (size_t)(n_cand * count));
    if (rc) return rc;
    return HICMI_OK;
}
}  // namespace

int hicmi_p2_score_window(hicmi_ctx* c, int64_t first, int64_t k, double* delta_out)
{
    if (!c || !delta_out) return fail(HICMI_EINVAL, "bad arguments");
    return window_batch(c, first, 1, k, delta_out);
}

// ---------------------------------------------------------------------------------------------------
// Whole decision steps in one call: fast scores of every candidate (device enumeration), short list
// of the candidates within 1e-9 of the step's best, literal re-scoring of the short list (cached by
// bin order under the current total), then the reference's first-strict-maximum scan (OG:349,359,
// 464,535).  The same logic as SubMatrix.first_strict_max in orderGenome.py, minus ~10 host round trips.
namespace {
const double kNearTop = 1e-9;

void use_total(hicmi_ctx* c, double total)
{
    if (!c->cache_valid || c->cache_total != total) { c->exact_cache.clear(); c->cache_total = total; c->cache_valid = true; }
}

// literal scores of rows (each n_used selection indices) under `total`, through the cache
int literal_scores(hicmi_ctx* c, const std::vector<std::vector<int32_t>>& rows, double total, std::vector<double>& out)
{
    out.assign(rows.size(), 0.0);
    if (rows.empty()) return HICMI_OK;
    const int64_t n_used = (int64_t)rows[0].size();
    std::vector<std::string> keys(rows.size());
    std::vector<size_t> todo;
    std::unordered_map<std::string, size_t> pending;
    for (size_t i = 0; i < rows.size(); i++) {
        keys[i].assign(reinterpret_cast<const char*>(rows[i].data()), rows[i].size() * sizeof(int32_t));
        if (c->exact_cache.count(keys[i]) || pending.count(keys[i])) continue;
        pending[keys[i]] = todo.size();
        todo.push_back(i);
    }
    if (!todo.empty()) {
        std::vector<double> vals(todo.size(), 0.0);
        if (n_used >= 2) {
            std::vector<int32_t> flat((size_t)n_used * todo.size());
            for (size_t t = 0; t < todo.size(); t++) memcpy(flat.data() + t * n_used, rows[todo[t]].data(), sizeof(int32_t) * (size_t)n_used);
            const int64_t n_cand = (int64_t)todo.size();
            int rc = ensure(c->d_perms, c->perms_cap, n_cand * n_used);
            if (rc) return rc;
            rc = ensure(c->d_scores, c->scores_cap, n_cand);
            if (rc) return rc;
            rc = ensure(c->d_T, c->t_cap, 2 * n_cand * n_used);
            if (rc) return rc;
            rc = upload(c, c->d_perms, flat.data(), sizeof(int32_t) * flat.size());
            if (rc) return rc;
            {
                Timed t(c, F_P2_EXACT, 8.0 * (double)n_cand * 0.5 * (double)n_used * (double)(n_used - 1));
                launch_p2_score_exact(c->dM2, c->ld2, c->d_perms, (int)n_cand, (int)n_used, total, c->d_T,
                                      c->d_T + n_cand * n_used, c->d_scores, c->stream);
            }
            HIPCHK(hipGetLastError());
            rc = download(c, vals.data(), c->d_scores, sizeof(double) * vals.size());
            if (rc) return rc;
        }
        for (size_t t = 0; t < todo.size(); t++) c->exact_cache[keys[todo[t]]] = vals[t];
    }
    for (size_t i = 0; i < rows.size(); i++) out[i] = c->exact_cache[keys[i]];
    return HICMI_OK;
}

// indices of the candidates whose fast score is within kNearTop of max(best fast, floor)
void short_list(const std::vector<double>& fast, double floor, std::vector<int64_t>& near)
{
    near.clear();
    bool any = false; double top = floor;
    for (double v : fast) if (std::isfinite(v)) { any = true; if (v > top) top = v; }
    if (!any) return;
    const double thr = top - std::fabs(top) * kNearTop;
    for (size_t i = 0; i < fast.size(); i++) if (std::isfinite(fast[i]) && fast[i] >= thr) near.push_back((int64_t)i);
}
}  // namespace

namespace {
// the decision of one window from its deltas; cur_fast (fast score of the current arrangement) is
// computed on first use when NaN
int decide_from_delta(hicmi_ctx* c, int64_t first, int64_t k, double total, double floor, double& cur_fast,
                      const double* delta, int64_t* pick_out, double* best_out, double* pick_fast_out)
{
    const int64_t S = (int64_t)c->h_arr_id.size();
    const int64_t n_ord = c->n_orders, n_ori = c->n_orients, n_cand = n_ord * n_ori;
    *pick_out = -1; *best_out = floor; *pick_fast_out = cur_fast;
    // candidate index of the current configuration: identity order + the window's current signs
    int64_t c0 = -1;
    for (int64_t r = 0; r < n_ori && c0 < 0; r++) {
        bool same = true;
        for (int64_t j = 0; j < k; j++) same = same && ((c->h_orients[(size_t)(r * k + j)] != 0) == (c->h_arr_rev[(size_t)(first + j)] != 0));
        if (same) c0 = r;
    }
    if (c0 < 0) return fail(HICMI_EINVAL, "current orientation not in the orientation table");
    std::vector<double> fast((size_t)n_cand);
    int rc;
    if (k == S) for (int64_t i = 0; i < n_cand; i++) fast[(size_t)i] = delta[(size_t)i] / total;
    else {
        if (std::isnan(cur_fast)) { rc = hicmi_p2_arrangement_score(c, total, &cur_fast); if (rc) return rc; }
        for (int64_t i = 0; i < n_cand; i++) fast[(size_t)i] = cur_fast + (delta[(size_t)i] - delta[(size_t)c0]) / total;
    }
    *pick_fast_out = cur_fast;
    std::vector<int64_t> near;
    short_list(fast, floor, near);
    if (near.empty()) return HICMI_OK;
    // The candidate that IS the current arrangement is near the top in every window (its fast score is the floor's twin)
    // and its bin order is the same in all of them: its literal score is worked out once per arrangement and total, not
    // through a 7 KB row and cache key per window (138 windows x 1-2 rounds per chromosome at 16k: ~3 ms of a 4.5 ms scan).
    double lit_c0 = 0.0; bool have_c0 = false;
    if (k != S) {
        for (int64_t cand : near) if (cand == c0) have_c0 = true;
        if (have_c0) {
            if (c->cur_lit_version != c->arr_version || c->cur_lit_total != total) {
                std::vector<std::vector<int32_t>> one(1, c->h_pos2sel);
                std::vector<double> v;
                rc = literal_scores(c, one, total, v);
                if (rc) return rc;
                c->cur_lit_version = c->arr_version; c->cur_lit_total = total; c->cur_lit_value = v[0];
            }
            lit_c0 = c->cur_lit_value;
            if (near.size() == 1) {                          // nothing but the arrangement itself: decided
                if (lit_c0 > floor) { *pick_out = c0; *best_out = lit_c0; *pick_fast_out = fast[(size_t)c0]; }
                return HICMI_OK;
            }
        }
    }
    const int p0 = c->h_arr_pos[(size_t)first], p1 = c->h_arr_pos[(size_t)(first + k)];
    std::vector<std::vector<int32_t>> rows(near.size());
    for (size_t q = 0; q < near.size(); q++) {
        const int64_t cand = near[q];
        const int8_t* ord = c->h_orders.data() + (cand / n_ori) * k;
        const uint8_t* ori = c->h_orients.data() + (cand % n_ori) * k;
        std::vector<int32_t>& row = rows[q];
        row.reserve((size_t)c->n_arr);
        row.insert(row.end(), c->h_pos2sel.begin(), c->h_pos2sel.begin() + p0);
        for (int64_t j = 0; j < k; j++) {
            const int32_t sc = c->h_arr_id[(size_t)(first + ord[j])];
            const int32_t st = c->h_scaf_start[(size_t)sc], ln = c->h_scaf_len[(size_t)sc];
            if (ori[j]) for (int32_t e = 0; e < ln; e++) row.push_back(st + ln - 1 - e);
            else        for (int32_t e = 0; e < ln; e++) row.push_back(st + e);
        }
        row.insert(row.end(), c->h_pos2sel.begin() + p1, c->h_pos2sel.end());
    }
    std::vector<double> lit;
    rc = literal_scores(c, rows, total, lit);
    if (rc) return rc;
    double best = floor; int64_t pick = -1;
    for (size_t q = 0; q < near.size(); q++) if (lit[q] > best) { best = lit[q]; pick = near[q]; }
    *pick_out = pick; *best_out = best;
    if (pick >= 0) *pick_fast_out = fast[(size_t)pick];
    return HICMI_OK;
}

int check_window_call(hicmi_ctx* c, int64_t first, int64_t k)
{
    const int64_t S = (int64_t)c->h_arr_id.size();
    if (c->n_arr < 1 || S < 1) return fail(HICMI_EINVAL, "hicmi_p2_set_arrangement has not run");
    if (k != c->tab_k || c->h_orders.empty()) return fail(HICMI_EINVAL, "hicmi_p2_window_tables has not been called for k = %lld", (long long)k);
    if (first < 0 || first + k > S) return fail(HICMI_EINVAL, "window out of range");
    for (int64_t j = 0; j < k; j++) if (c->h_orders[(size_t)j] != j) return fail(HICMI_EINVAL, "orders[0] must be the identity");
    return HICMI_OK;
}
}  // namespace

int hicmi_p2_decide_window(hicmi_ctx* c, int64_t first, int64_t k, double total, double floor, double cur_fast,
                           int64_t* pick_out, double* best_out, double* pick_fast_out)
{
    if (!c || !pick_out || !best_out || !pick_fast_out) return fail(HICMI_EINVAL, "NULL argument");
    int rc = check_window_call(c, first, k);
    if (rc) return rc;
    use_total(c, total);
    std::vector<double> delta((size_t)(c->n_orders * c->n_orients));
    rc = window_batch(c, first, 1, k, delta.data());
    if (rc) return rc;
    return decide_from_delta(c, first, k, total, floor, cur_fast, delta.data(), pick_out, best_out, pick_fast_out);
}

int hicmi_p2_decide_insertion(hicmi_ctx* c, const int32_t* ids, const uint8_t* rev, int64_t S, int32_t new_id,
                              int32_t new_rev_now, int64_t* gap_out, int32_t* rev_out, double* best_out)
{
    if (!c || !ids || !rev || !gap_out || !rev_out || !best_out || S < 1) return fail(HICMI_EINVAL, "bad arguments");
    if (c->n_scaf < 1) return fail(HICMI_EINVAL, "hicmi_p2_layout has not run");
    if (new_id < 0 || new_id >= c->n_scaf) return fail(HICMI_EINVAL, "new scaffold out of range");
    *gap_out = -1; *rev_out = 0; *best_out = 0.0;
    // total of the sub-matrix in the order "ordered scaffolds, then the new one" (OG:484-487 -> OG:343)
    std::vector<int32_t> ids2(ids, ids + S); ids2.push_back(new_id);
    std::vector<uint8_t> rev2(rev, rev + S); rev2.push_back(new_rev_now ? 1 : 0);
    int rc = hicmi_p2_set_arrangement(c, ids2.data(), rev2.data(), S + 1);
    if (rc) return rc;
    double total = 0.0;
    rc = hicmi_p2_arrangement_total(c, &total);
    if (rc) return rc;
    use_total(c, total);
    rc = hicmi_p2_set_arrangement(c, ids, rev, S);
    if (rc) return rc;
    const int64_t gaps = S + 1;
    std::vector<double> by_gap_rev((size_t)(2 * gaps));
    rc = hicmi_p2_score_insertions(c, new_id, total, by_gap_rev.data());
    if (rc) return rc;
    // enumeration order of OG:344-365: gap i tests the current orientation, then the flipped one, and the
    // scaffold stays flipped for the next gap
    std::vector<double> fast((size_t)(2 * gaps));
    std::vector<int32_t> tag_rev((size_t)(2 * gaps));
    int32_t o = new_rev_now ? 1 : 0;
    for (int64_t i = 0; i < gaps; i++) {
        tag_rev[(size_t)(2 * i)] = o; tag_rev[(size_t)(2 * i + 1)] = o ^ 1;
        fast[(size_t)(2 * i)] = by_gap_rev[(size_t)(2 * i + o)];
        fast[(size_t)(2 * i + 1)] = by_gap_rev[(size_t)(2 * i + (o ^ 1))];
        o ^= 1;
    }
    std::vector<int64_t> near;
    short_list(fast, 0.0, near);
    if (near.empty()) return HICMI_OK;
    const int32_t st = c->h_scaf_start[(size_t)new_id], ln = c->h_scaf_len[(size_t)new_id];
    std::vector<std::vector<int32_t>> rows(near.size());
    for (size_t q = 0; q < near.size(); q++) {
        const int64_t g = near[q] / 2;
        const int32_t r = tag_rev[(size_t)near[q]];
        const int P = c->h_arr_pos[(size_t)g];
        std::vector<int32_t>& row = rows[q];
        row.reserve((size_t)(c->n_arr + ln));
        row.insert(row.end(), c->h_pos2sel.begin(), c->h_pos2sel.begin() + P);
        if (r) for (int32_t e = 0; e < ln; e++) row.push_back(st + ln - 1 - e);
        else   for (int32_t e = 0; e < ln; e++) row.push_back(st + e);
        row.insert(row.end(), c->h_pos2sel.begin() + P, c->h_pos2sel.end());
    }
    std::vector<double> lit;
    rc = literal_scores(c, rows, total, lit);
    if (rc) return rc;
    double best = 0.0; int64_t pick = -1;
    for (size_t q = 0; q < near.size(); q++) if (lit[q] > best) { best = lit[q]; pick = near[q]; }
    if (pick >= 0) { *gap_out = pick / 2; *rev_out = tag_rev[(size_t)pick]; *best_out = best; }
    return HICMI_OK;
}


// Whole loops of the search, so that a chromosome costs a handful of host calls (several chromosomes
// run concurrently from host threads, each on its own context).
namespace {
void apply_insertion(int32_t* ids, uint8_t* rev, int64_t& S, int64_t gap, int32_t new_id, int32_t r)
{
    for (int64_t j = S; j > gap; j--) { ids[j] = ids[j - 1]; rev[j] = rev[j - 1]; }
    ids[gap] = new_id; rev[gap] = (uint8_t)(r ? 1 : 0);
    S++;
}

struct InsJob {
    hicmi_ctx* c; int32_t* ids; uint8_t* rev; int64_t S; const int32_t* new_ids; int64_t n_new;
    int64_t t = 0; double best = 0.0; bool device_ok = true;
};

// Queue the remaining insertions of every job in lock step on `lead`'s stream, with the decisions taken on
// the device (k_part2_insert.hip), and synchronise once.  Decided steps are applied to ids/rev/S/t of each
// job; a job stops early at a step the device declined (short list longer than the cap), which the host
// has to decide.
int queue_insertions(hicmi_ctx* lead, const std::vector<InsJob*>& jobs)
{
    const int nj = (int)jobs.size();
    if (nj == 0) return HICMI_OK;
    const int NB = kBaseSlabs;
    static const int max_c = getenv("HICMI_P2_INS_MAXC") ? atoi(getenv("HICMI_P2_INS_MAXC")) : INS_MAXC;   // tests: force host steps
    int64_t steps_max = 0;
    std::vector<size_t> blob_off((size_t)nj);
    size_t blob_bytes = 0;
    for (int j = 0; j < nj; j++) {
        InsJob& job = *jobs[(size_t)j];
        hicmi_ctx* c = job.c;
        int rc = hicmi_p2_set_arrangement(c, job.ids, job.rev, job.S);
        if (rc) return rc;
        if (c != lead) HIPCHK(sync_stream(c));             // the queue runs on lead's stream
        const int64_t n_steps = job.n_new - job.t;
        int64_t n_max = c->n_arr;
        for (int64_t t = 0; t < n_steps; t++) n_max += c->h_scaf_len[(size_t)job.new_ids[job.t + t]];
        if (n_max * (int64_t)sizeof(int32_t) > 160 * 1024) return fail(HICMI_EUNSUPPORTED, "candidate longer than 40960 bins");
        const int64_t S_max = job.S + n_steps;
        rc = ensure(c->d_arr_packed2, c->arr2_cap, 3 * std::max<int64_t>(S_max, c->n_scaf) + 2);
        if (rc) return rc;
        rc = ensure(c->d_pos2sel2, c->pos2_cap, c->n2);
        if (rc) return rc;
        rc = ensure(c->d_ins_T, c->ins_t_cap, (1 + 2 * (int64_t)INS_MAXC) * n_max);
        if (rc) return rc;
        rc = ensure(c->d_ins_partial, c->ins_partial_cap, NB + n_max + 2 * (S_max + 1));
        if (rc) return rc;
        steps_max = std::max(steps_max, n_steps);
        blob_off[(size_t)j] = blob_bytes;
        blob_bytes += (sizeof(InsState) + sizeof(InsLog) * (size_t)n_steps + 15) & ~(size_t)15;
    }
    HIPCHK(hipSetDevice(lead->device));
    int rc = ensure(lead->d_ins_blob, lead->ins_blob_cap, (int64_t)blob_bytes);
    if (rc) return rc;
    rc = ensure(lead->d_ins_steps, lead->ins_steps_cap, steps_max * nj);
    if (rc) return rc;
    // one record per (step, job); sizes per step for the launch grids
    std::vector<InsStep> table((size_t)(steps_max * nj));
    std::vector<int> max_n_used((size_t)steps_max, 0), max_S((size_t)steps_max, 0), max_n_arr((size_t)steps_max, 0);
    double algo = 0.0;
    for (int j = 0; j < nj; j++) {
        InsJob& job = *jobs[(size_t)j];
        hicmi_ctx* c = job.c;
        const int64_t n_steps = job.n_new - job.t;
        int64_t n_max = c->n_arr;
        for (int64_t t = 0; t < n_steps; t++) n_max += c->h_scaf_len[(size_t)job.new_ids[job.t + t]];
        InsState* st = reinterpret_cast<InsState*>(lead->d_ins_blob + blob_off[(size_t)j]);
        InsLog* log = reinterpret_cast<InsLog*>(lead->d_ins_blob + blob_off[(size_t)j] + sizeof(InsState));
        int32_t* packed[2] = {c->d_arr_packed, c->d_arr_packed2};
        int32_t* pos2sel[2] = {c->d_pos2sel, c->d_pos2sel2};
        int64_t n_arr = c->n_arr;
        for (int64_t t = 0; t < steps_max; t++) {
            InsStep& d = table[(size_t)(t * nj + j)];
            memset(&d, 0, sizeof(d));
            d.st = st;
            if (t >= n_steps) continue;                       // this chromosome has finished: inactive record
            const int cur = (int)(t & 1), nxt = cur ^ 1;
            const int32_t nid = job.new_ids[job.t + t];
            const int L = c->h_scaf_len[(size_t)nid];
            d.M2 = c->dM2; d.H = c->d_H; d.ld2 = c->ld2;
            d.pos_cur = pos2sel[cur]; d.pos_nxt = pos2sel[nxt]; d.packed_cur = packed[cur]; d.packed_nxt = packed[nxt];
            d.T_total = c->d_ins_T; d.T_cand = c->d_ins_T + n_max; d.work = c->d_ins_T + (1 + (int64_t)INS_MAXC) * n_max;
            d.partial = c->d_ins_partial; d.log = log + t;
            d.n_arr = (int32_t)n_arr; d.S = (int32_t)(job.S + t); d.L = L; d.new_start = c->h_scaf_start[(size_t)nid];
            d.new_id = nid; d.active = 1; d.step = (int32_t)t; d.last = t == n_steps - 1;
            max_n_used[(size_t)t] = std::max(max_n_used[(size_t)t], (int)(n_arr + L));
            max_S[(size_t)t] = std::max(max_S[(size_t)t], d.S);
            max_n_arr[(size_t)t] = std::max(max_n_arr[(size_t)t], (int)n_arr);
            const double nn = (double)n_arr, Ld = (double)L, sd = (double)d.S;
            algo += 8.0 * (0.5 * nn * (nn - 1.0) + nn * nn + 2.0 * (sd + 1.0) * Ld * nn) + 4.0 * (nn + Ld) * (nn + Ld);
            n_arr += L;
        }
    }
    rc = upload(lead, lead->d_ins_steps, table.data(), sizeof(InsStep) * table.size());
    if (rc) return rc;
    {
        Timed timed(lead, F_P2_INSERT, algo);
        launch_insb_reset(lead->d_ins_steps, nj, lead->stream);
        for (int64_t t = 0; t < steps_max; t++) {
            const InsStep* st_t = lead->d_ins_steps + t * nj;
            const int nu = max_n_used[(size_t)t];
            launch_insb_fast(st_t, nj, max_S[(size_t)t], max_n_arr[(size_t)t], NB, lead->stream);
            launch_insb_shortlist(st_t, nj, max_S[(size_t)t], max_n_arr[(size_t)t], NB, kNearTop, max_c, lead->stream);
            launch_insb_diag_cand(st_t, nj, nu, lead->stream);
            launch_insb_cost(st_t, nj, nu, lead->stream);
            launch_insb_apply(st_t, nj, nu, lead->stream);
        }
    }
    HIPCHK(hipGetLastError());
    std::vector<unsigned char> blob(blob_bytes);
    rc = download(lead, blob.data(), lead->d_ins_blob, blob_bytes);
    if (rc) return rc;
    if (getenv("HICMI_PART2_PROFILE")) {                   // how many lock steps needed a literal tie-break at all
        int64_t hist[5] = {0, 0, 0, 0, 0}, all_direct = 0;       // [0] = taken directly (one candidate near the top)
        for (int64_t t = 0; t < steps_max; t++) {
            int mx = -1;
            for (int j = 0; j < nj; j++) {
                if (t >= jobs[(size_t)j]->n_new - jobs[(size_t)j]->t) continue;
                const InsLog* hl = reinterpret_cast<const InsLog*>(blob.data() + blob_off[(size_t)j] + sizeof(InsState));
                const int ns = hl[t].n_short;
                hist[ns < 0 ? 0 : (ns > 3 ? 4 : ns + 1)]++;
                mx = std::max(mx, ns);
            }
            if (mx < 0) all_direct++;
        }
        fprintf(stderr, "[hicmi] insertion short lists: direct %lld, literal with 0:%lld 1:%lld 2:%lld 3+:%lld candidates; "
                        "lock steps without any literal pass: %lld of %lld\n",
                (long long)hist[0], (long long)hist[1], (long long)hist[2], (long long)hist[3], (long long)hist[4],
                (long long)all_direct, (long long)steps_max);
    }
    for (int j = 0; j < nj; j++) {
        InsJob& job = *jobs[(size_t)j];
        const int64_t n_steps = job.n_new - job.t;
        const InsState* hst = reinterpret_cast<const InsState*>(blob.data() + blob_off[(size_t)j]);
        const InsLog* hlog = reinterpret_cast<const InsLog*>(blob.data() + blob_off[(size_t)j] + sizeof(InsState));
        const int64_t done = hst->fail >= 0 ? std::min<int64_t>(hst->fail, n_steps) : n_steps;
        for (int64_t t = 0; t < done; t++) {
            if (hlog[t].gap < 0 || hlog[t].gap > job.S) return fail(HICMI_ESTATE, "insertion log out of range");
            apply_insertion(job.ids, job.rev, job.S, hlog[t].gap, job.new_ids[job.t + t], hlog[t].rev);
            job.best = hlog[t].best;
        }
        job.t += done;
        // the device buffers hold a different arrangement from the host mirrors now
        hicmi_ctx* c = job.c;
        c->n_arr = 0; c->h_arr_id.clear(); c->h_arr_rev.clear(); c->h_arr_pos.clear(); c->h_pos2sel.clear();
    }
    return HICMI_OK;
}

int check_insert_job(const InsJob& job, int64_t S0)
{
    hicmi_ctx* c = job.c;
    if (!c || !job.ids || !job.rev || !job.new_ids || S0 < 1 || job.n_new < 1) return fail(HICMI_EINVAL, "bad arguments");
    if (c->n_scaf < 1) return fail(HICMI_EINVAL, "hicmi_p2_layout has not run");
    std::vector<uint8_t> used((size_t)c->n_scaf, 0);
    for (int64_t j = 0; j < S0 + job.n_new; j++) {
        const int32_t v = j < S0 ? job.ids[j] : job.new_ids[j - S0];
        if (v < 0 || v >= c->n_scaf || used[(size_t)v]) return fail(HICMI_EINVAL, "scaffolds must be distinct members of the layout");
        used[(size_t)v] = 1;
    }
    return HICMI_OK;
}

int host_insertion_step(InsJob& job)
{
    int64_t gap = -1; int32_t r = 0;
    int rc = hicmi_p2_decide_insertion(job.c, job.ids, job.rev, job.S, job.new_ids[job.t], 0, &gap, &r, &job.best);
    if (rc) return rc;
    if (gap < 0) { gap = 0; r = 0; job.best = 0.0; }
    apply_insertion(job.ids, job.rev, job.S, gap, job.new_ids[job.t], r);
    job.t++;
    return HICMI_OK;
}

int run_insert_jobs(std::vector<InsJob>& all)
{
    static const bool host_only = getenv("HICMI_P2_HOST_INSERT") != nullptr;     // A/B switch: every step decided by the host
    for (InsJob& job : all) {
        int rc = check_insert_job(job, job.S);
        if (rc) return rc;
        if (job.S + job.n_new + 1 > 8192) job.device_ok = false;   // prefix table of k_insb_shortlist; the host path has no such limit
    }
    while (true) {
        std::vector<InsJob*> todo;
        for (InsJob& job : all) if (job.t < job.n_new && job.device_ok && !host_only) todo.push_back(&job);
        if (!todo.empty()) {
            int rc = queue_insertions(todo[0]->c, todo);
            if (rc) return rc;
        }
        bool pending = false;
        for (InsJob& job : all) {
            if (job.t >= job.n_new) continue;
            // the device declined this step (or may not be used): the host decides it
            do {
                int rc = host_insertion_step(job);
                if (rc) return rc;
            } while (job.t < job.n_new && (host_only || !job.device_ok));
            if (job.t < job.n_new) pending = true;
        }
        if (!pending) break;
    }
    return HICMI_OK;
}
}  // namespace

int hicmi_p2_insert_all(hicmi_ctx* c, int32_t* ids, uint8_t* rev, int64_t S0, const int32_t* new_ids, int64_t n_new,
                        double* best_out)
{
    // orderRemainderScaffolds (OG:475-493) for n_new >= 1 scaffolds pulled in order: ids/rev hold S0 entries
    // on entry and S0 + n_new on return (capacity is the caller's).  Each new scaffold enters in '+'
    // orientation (it has never been flipped, OG:265) and leaves checkAllScores in the winning
    // orientation - '+' when nothing scored above 0 (OG:341, 367-368) - at the winning gap (0 by default).
    if (!c || !best_out) return fail(HICMI_EINVAL, "bad arguments");
    std::vector<InsJob> jobs(1);
    jobs[0].c = c; jobs[0].ids = ids; jobs[0].rev = rev; jobs[0].S = S0; jobs[0].new_ids = new_ids; jobs[0].n_new = n_new;
    int rc = run_insert_jobs(jobs);
    if (rc) return rc;
    *best_out = jobs[0].best;
    return HICMI_OK;
}

int hicmi_p2_insert_all_multi(int64_t n_jobs, hicmi_ctx* const* ctxs, int32_t* const* ids, uint8_t* const* rev,
                              const int64_t* S0, const int32_t* const* new_ids, const int64_t* n_new, double* best_out)
{
    // the same for several chromosomes (one context each, all on one device), advanced in lock step: one
    // launch per kernel and step serves every chromosome that still has scaffolds to place
    if (n_jobs < 1 || !ctxs || !ids || !rev || !S0 || !new_ids || !n_new || !best_out) return fail(HICMI_EINVAL, "bad arguments");
    std::vector<InsJob> jobs((size_t)n_jobs);
    for (int64_t j = 0; j < n_jobs; j++) {
        InsJob& job = jobs[(size_t)j];
        job.c = ctxs[j]; job.ids = ids[j]; job.rev = rev[j]; job.S = S0[j]; job.new_ids = new_ids[j]; job.n_new = n_new[j];
        if (!job.c || job.c->device != ctxs[0]->device) return fail(HICMI_EINVAL, "contexts must share one device");
        for (int64_t q = 0; q < j; q++) if (ctxs[q] == ctxs[j]) return fail(HICMI_EINVAL, "one context per chromosome");
    }
    int rc = run_insert_jobs(jobs);
    if (rc) return rc;
    for (int64_t j = 0; j < n_jobs; j++) best_out[j] = jobs[(size_t)j].best;
    return HICMI_OK;
}

int hicmi_p2_scan_pass(hicmi_ctx* c, int32_t* ids, uint8_t* rev, int64_t S, int64_t k, double total, double* best_io,
                       double* cur_fast_io, int32_t* improved_out)
{
    // one round of scanOrdering (OG:513-541): windows first = 0 .. S-k, each applied before the next.
    // Until a window improves, all of them see the same arrangement, so they are scored in batches of
    // up to 32 windows per launch pair; after an improvement the remaining windows are scored again
    // against the new arrangement.
    if (!c || !ids || !rev || !best_io || !cur_fast_io || !improved_out || S < 1 || k < 1 || k > S)
        return fail(HICMI_EINVAL, "bad arguments");
    *improved_out = 0;
    const int64_t n_ori = c->n_orients, n_cand = c->n_orders * c->n_orients;
    int rc = hicmi_p2_set_arrangement(c, ids, rev, S);
    if (rc) return rc;
    rc = check_window_call(c, 0, k);
    if (rc) return rc;
    use_total(c, total);
    const int64_t last = S - k;
    std::vector<double> delta;
    int64_t first = 0;
    // windows scored per launch pair: 8, doubling while no window improves (an improvement throws the rest of the batch
    // away); a round that follows a round with at most one improvement starts at 32 (hicmi_p2_scan_all: the late rounds
    // of a chromosome change little, and a batch costs a launch pair + a synchronisation whatever its size)
    int64_t batch = c->scan_first_batch > 0 ? c->scan_first_batch : 8;
    int64_t n_improved = 0;
    while (first <= last) {
        const int64_t count = std::min<int64_t>(batch, last - first + 1);
        delta.resize((size_t)(count * n_cand));
        rc = window_batch(c, first, count, k, delta.data());
        if (rc) return rc;
        bool applied = false;
        for (int64_t wdx = 0; wdx < count && !applied; wdx++) {
            int64_t pick = -1; double best = *best_io, pf = *cur_fast_io;
            double cf = *cur_fast_io;
            rc = decide_from_delta(c, first + wdx, k, total, *best_io, cf, delta.data() + wdx * n_cand, &pick, &best, &pf);
            if (rc) return rc;
            *cur_fast_io = pf;
            if (pick < 0) continue;
            *best_io = best; *improved_out = 1;
            const int64_t f = first + wdx;
            const int8_t* ord = c->h_orders.data() + (pick / n_ori) * k;
            const uint8_t* ori = c->h_orients.data() + (pick % n_ori) * k;
            int32_t wid[8];
            for (int64_t j = 0; j < k; j++) wid[j] = ids[f + ord[j]];
            for (int64_t j = 0; j < k; j++) { ids[f + j] = wid[j]; rev[f + j] = ori[j] ? 1 : 0; }
            rc = hicmi_p2_set_arrangement(c, ids, rev, S);
            if (rc) return rc;
            first = f + 1;
            applied = true;
            n_improved++;
            batch = 8;
        }
        if (!applied) { first += count; batch = std::min<int64_t>(batch * 2, 32); }
    }
    c->scan_last_improved = n_improved;
    return HICMI_OK;
}

int hicmi_p2_scan_all(hicmi_ctx* c, int32_t* ids, uint8_t* rev, int64_t S, int64_t k, double total, double* best_io,
                      double* cur_fast_io, int64_t* rounds_out)
{
    // scanOrdering's outer loop (OG:509-547) as ONE call: a chromosome's ~20 rounds are ~20 returns to the interpreter
    // otherwise, each of which waits for the interpreter lock behind the other chromosomes' threads
    if (!rounds_out) return fail(HICMI_EINVAL, "bad arguments");
    *rounds_out = 0;
    c->scan_first_batch = 0;
    for (;;) {
        int32_t improved = 0;
        int rc = hicmi_p2_scan_pass(c, ids, rev, S, k, total, best_io, cur_fast_io, &improved);
        c->scan_first_batch = (rc == 0 && c->scan_last_improved <= 1) ? 32 : 0;
        if (rc || !improved) { c->scan_first_batch = 0; }
        if (rc) return rc;
        ++*rounds_out;
        if (!improved) return HICMI_OK;
        if (*rounds_out > 100000) return fail(HICMI_ESTATE, "scanOrdering does not converge");
    }
}

// ---- plot support (plotContactMaps.py:15-91) --------------------------------------------------------
namespace {
int plot_prepare(hicmi_ctx* c, int kind, const int32_t* order, int64_t n_sel, int32_t** d_order_out)
{
    if (!c->dC) return fail(HICMI_EINVAL, "no contact matrix set");
    if (kind < 0 || kind > 2) return fail(HICMI_EINVAL, "kind must be 0 (contacts), 1 (distance) or 2 (similarity)");
    if (n_sel < 1 || (!order && n_sel != c->n)) return fail(HICMI_EINVAL, "n_sel must be the matrix size when no order is given");
    HIPCHK(hipSetDevice(c->device));
    if (kind != 0 && !c->have_sums) { int rc = compute_sums(c); if (rc) return rc; }
    *d_order_out = nullptr;
    if (order) {
        for (int64_t i = 0; i < n_sel; i++) if (order[i] < 0 || order[i] >= c->n) return fail(HICMI_EINVAL, "order entry out of range");
        int rc = ensure(c->d_plot_order, c->plot_order_cap, n_sel);
        if (rc) return rc;
        rc = upload(c, c->d_plot_order, order, sizeof(int32_t) * (size_t)n_sel);
        if (rc) return rc;
        *d_order_out = c->d_plot_order;
    }
    return HICMI_OK;
}
}  // namespace

int hicmi_plot_percentiles(hicmi_ctx* c, int kind, const int32_t* order, int64_t n_sel, const double* q, int64_t n_q,
                           double* out)
{
    // numpy.percentile(a, q) (method "linear") over the n_sel x n_sel cells: virtual index (N-1)*q/100, the two
    // neighbouring order statistics selected exactly on the device, numpy's _lerp on the host
    if (!c || !q || !out || n_q < 1) return fail(HICMI_EINVAL, "bad arguments");
    for (int64_t i = 0; i < n_q; i++) if (!(q[i] >= 0.0 && q[i] <= 100.0)) return fail(HICMI_EINVAL, "percentiles must be in [0, 100]");
    int32_t* d_order = nullptr;
    int rc = plot_prepare(c, kind, order, n_sel, &d_order);
    if (rc) return rc;
    rc = ensure(c->d_plot_work, c->plot_work_cap, (int64_t)(plot_select_state_bytes() + plot_select_hist_bytes()));
    if (rc) return rc;
    SelectState* d_state = reinterpret_cast<SelectState*>(c->d_plot_work);
    unsigned int* d_hist = reinterpret_cast<unsigned int*>(c->d_plot_work + plot_select_state_bytes());
    const double N = (double)n_sel * (double)n_sel;
    const int per_batch = plot_select_max_targets() / 2;
    std::vector<unsigned char> host(plot_select_state_bytes());
    for (int64_t q0 = 0; q0 < n_q; q0 += per_batch) {
        const int nb = (int)std::min<int64_t>(per_batch, n_q - q0);
        unsigned long long ranks[8];
        double frac[4];
        for (int t = 0; t < nb; t++) {
            const double virt = (N - 1.0) * (q[q0 + t] / 100.0);        // numpy: quantile * (n - 1)
            double lo = std::floor(virt);
            if (lo > N - 1.0) lo = N - 1.0;
            const double hi = std::min(lo + 1.0, N - 1.0);
            ranks[2 * t] = (unsigned long long)lo; ranks[2 * t + 1] = (unsigned long long)hi;
            frac[t] = virt - lo;
        }
        plot_select_fill(host.data(), ranks, 2 * nb);
        rc = upload(c, d_state, host.data(), host.size());
        if (rc) return rc;
        HIPCHK(hipMemsetAsync(d_hist, 0, plot_select_hist_bytes(), c->stream));
        {
            Timed t(c, F_PLOT, 6.0 * 8.0 * N);
            launch_plot_select(c->dC, c->ldc, c->d_np, c->d_seq, kind, d_order, (int)n_sel, 2 * nb, d_state, d_hist, c->stream);
        }
        HIPCHK(hipGetLastError());
        rc = download(c, host.data(), d_state, host.size());
        if (rc) return rc;
        for (int t = 0; t < nb; t++) {
            const double a = plot_select_value(host.data(), 2 * t), b = plot_select_value(host.data(), 2 * t + 1), g = frac[t];
            const double diff = b - a;                                  // numpy.lib._function_base_impl._lerp
            double v = a + diff * g;
            if (g >= 0.5) v = b - diff * (1.0 - g);
            if (diff == 0.0) v = a;
            out[q0 + t] = v;
        }
    }
    return HICMI_OK;
}

int hicmi_plot_downsample(hicmi_ctx* c, int kind, const int32_t* order, int64_t n_sel, int64_t px, double* out)
{
    if (!c || !out || px < 1 || px > n_sel) return fail(HICMI_EINVAL, "bad arguments (1 <= px <= n_sel)");
    int32_t* d_order = nullptr;
    int rc = plot_prepare(c, kind, order, n_sel, &d_order);
    if (rc) return rc;
    rc = ensure(c->d_plot_img, c->plot_img_cap, px * px);
    if (rc) return rc;
    {
        Timed t(c, F_PLOT, 8.0 * (double)n_sel * (double)n_sel);
        launch_plot_downsample(c->dC, c->ldc, c->d_np, c->d_seq, kind, d_order, (int)n_sel, (int)px, c->d_plot_img, c->stream);
    }
    HIPCHK(hipGetLastError());
    return download(c, out, c->d_plot_img, sizeof(double) * (size_t)(px * px));
}

// ---------------------------------------------------------------------------------------------------
int hicmi_timing_reset(hicmi_ctx* c)
{
    if (!c) return fail(HICMI_EINVAL, "NULL context");
    int rc = resolve_timing(c);
    if (rc) return rc;
    for (int f = 0; f < F_COUNT; f++) { c->ms[f] = 0; c->launches[f] = 0; c->bytes[f] = 0; }
    c->nn_scans = c->nn_scan_cols = c->nn_cache_hits = c->nn_merges = 0; c->nn_retries = 0;
    return HICMI_OK;
}

int hicmi_timing_enable(hicmi_ctx* c, int on)
{
    if (!c) return fail(HICMI_EINVAL, "NULL context");
    int rc = resolve_timing(c);
    if (rc) return rc;
    c->timing = on < 0 ? 0 : (on > 2 ? 1 : on);
    return HICMI_OK;
}

int hicmi_timing_get(hicmi_ctx* c, char* names_out, int64_t names_cap, double* ms_out, int64_t* launches_out,
                     double* bytes_out, int64_t cap, int64_t* count_out)
{
    if (!c || !names_out || !ms_out || !launches_out || !bytes_out || !count_out) return fail(HICMI_EINVAL, "NULL argument");
    if (cap < F_COUNT) return fail(HICMI_EINVAL, "need room for %d families", (int)F_COUNT);
    int rc = resolve_timing(c);
    if (rc) return rc;
    std::string names;
    for (int f = 0; f < F_COUNT; f++) {
        if (f) names += ";";
        names += kFamilyNames[f];
        ms_out[f] = c->ms[f]; launches_out[f] = c->launches[f]; bytes_out[f] = c->bytes[f];
    }
    if ((int64_t)names.size() + 1 > names_cap) return fail(HICMI_EINVAL, "names buffer too small");
    memcpy(names_out, names.c_str(), names.size() + 1);
    *count_out = F_COUNT;
    return HICMI_OK;
}

}  // extern "C"
