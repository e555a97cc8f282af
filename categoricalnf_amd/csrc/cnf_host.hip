// Host-side plumbing of libcnf_hip.so: status strings, launch checks, row tiling.
#include "cnf_common.h"

#include <algorithm>
#include <atomic>
#include <string.h>
#include <vector>

namespace cnf {

static thread_local char g_err[512] = "";
// defaults from the interleaved A/B sweep on MI355X (tools/sweep_affine.py, profiles/r01_sweep_affine.txt).
// Process-wide knobs, relaxed atomics: a host thread per device (nn.DataParallel replicas) may launch while another thread
// sets one — it then sees the old or the new value, never a torn one.  They are NOT per device or per thread: set them
// before use.
static std::atomic<int> g_tile_chunks{128};
static std::atomic<int> g_unroll{2};
static std::atomic<int> g_math{1};
static std::atomic<int> g_inverse{1};
static std::atomic<int> g_mix_tile{128};

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int launch_status(const char* what) {
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) return CNF_OK;
    set_error("%s: %s", what, hipGetErrorString(e));
    return CNF_ERR_LAUNCH;
}

int tile_chunks_target() { return g_tile_chunks.load(std::memory_order_relaxed); }
int unroll_target() { return g_unroll.load(std::memory_order_relaxed); }
int math_mode() { return g_math.load(std::memory_order_relaxed); }
int inverse_mode() { return g_inverse.load(std::memory_order_relaxed); }
int mixture_tile_items() { return g_mix_tile.load(std::memory_order_relaxed); }

// ---- kernel timing: event pairs bound to the dispatch packets of armed launches (host thread local) ----------
struct ProfState {
    std::vector<hipEvent_t> start, stop;
    int used = 0;       // pairs handed out since the last collect
    int armed = 0;      // launches still to be timed
};
static thread_local ProfState g_prof;
constexpr int kMaxProfPairs = 8192;

__global__ void cnf_zero_fill_kernel(uint32_t* p, size_t words) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (size_t)gridDim.x * blockDim.x) p[i] = 0u;
}

bool prof_take(hipEvent_t* start, hipEvent_t* stop) {
    ProfState& p = g_prof;
    if (p.armed <= 0) return false;
    --p.armed;
    if (p.used >= kMaxProfPairs) return false;
    if (p.used == (int)p.start.size()) {
        hipEvent_t a, b;
        if (hipEventCreate(&a) != hipSuccess) return false;
        if (hipEventCreate(&b) != hipSuccess) {
            (void)hipEventDestroy(a);
            return false;
        }
        p.start.push_back(a);
        p.stop.push_back(b);
    }
    *start = p.start[p.used];
    *stop = p.stop[p.used];
    ++p.used;
    return true;
}

RowTiling make_row_tiling(int B, int L, int force_vec, int target_chunks, int gran) {
    RowTiling t;
    t.B = B;
    t.L = L;
    t.vec = force_vec ? force_vec : (L % 4 == 0 ? 4 : (L % 2 == 0 ? 2 : 1));
    t.cpr = L / t.vec;
    const int target = std::min(std::max(target_chunks, kWave), kMaxTileChunks);
    int best = 1;
    if (t.cpr < target) {
        // rows per wave: keep the 64 lanes busy (chunks close to a multiple of 64), prefer more rows
        double best_eff = -1.0;
        const int rmax = std::max(1, std::min(target / t.cpr, std::max(1, B)));
        for (int r = 1; r <= rmax; ++r) {
            const int ch = r * t.cpr;
            // `gran` chunks are walked per iteration (64 lanes x chunks in flight per lane)
            const double eff = (double)ch / (double)(((ch + gran - 1) / gran) * gran);
            if (eff >= best_eff - 1e-9) {
                best_eff = std::max(eff, best_eff);
                best = r;
            }
        }
    }
    t.rw = best;
    int p2 = 1;
    while (p2 < t.rw && p2 < kWave) p2 <<= 1;
    t.p2 = p2;
    // few, longish rows (training-sized batches): one workgroup (4 waves) per row instead of one wave per
    // one-or-more rows, so that B rows still put 4B waves on the chip
    t.bpr = (B < 2048 && t.cpr >= kWave) ? 1 : 0;
    if (t.bpr) {
        t.rw = 1;
        t.p2 = 1;
    }
    t.ntiles = ((long)B + t.rw - 1) / t.rw;
    t.div_cpr = make_fastdiv((uint32_t)t.cpr);
    return t;
}

}  // namespace cnf

extern "C" {

int cnf_abi_version(void) { return 1; }

const char* cnf_last_error(void) { return cnf::g_err; }

void cnf_set_tile_chunks(int chunks) {
    if (chunks >= 64 && chunks <= cnf::kMaxTileChunks) cnf::g_tile_chunks.store(chunks, std::memory_order_relaxed);
}

void cnf_set_unroll(int u) {
    if (u == 0 || u == 1 || u == 2 || u == 3 || u == 4) cnf::g_unroll.store(u, std::memory_order_relaxed);
}

void cnf_set_math_mode(int mode) {
    if (mode == 0 || mode == 1) cnf::g_math.store(mode, std::memory_order_relaxed);
}

void cnf_set_mixture_tile(int items) {
    if (items >= 64 && items <= cnf::kMaxTileChunks) cnf::g_mix_tile.store(items, std::memory_order_relaxed);
}

void cnf_set_inverse_mode(int mode) {
    if (mode == 0 || mode == 1) cnf::g_inverse.store(mode, std::memory_order_relaxed);
}

int cnf_prof_arm(int launches) {
    cnf::g_prof.armed = launches > 0 ? launches : 0;
    return CNF_OK;
}

int cnf_prof_collect(float* ms_out, int capacity) {
    cnf::ProfState& p = cnf::g_prof;
    int n = 0;
    for (int i = 0; i < p.used; ++i) {
        if (hipEventSynchronize(p.stop[i]) != hipSuccess) continue;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.start[i], p.stop[i]) != hipSuccess) continue;
        if (ms_out && n < capacity) ms_out[n] = ms;
        ++n;
    }
    p.used = 0;
    p.armed = 0;
    (void)hipGetLastError();
    return n < capacity ? n : capacity;
}

}  // extern "C"
