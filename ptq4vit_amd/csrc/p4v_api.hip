// Host side of the C ABI declared in include/ptq4vit_hip.h.
//
// Each *_calibrate entry point enqueues the complete calibration_step2() of one module on the
// caller's stream: interval initialisation, candidate tables, and for every search round the
// pack -> sweep -> finish -> select kernel chain for both operands.  Intervals live in device
// memory from start to end, so there is no host round trip inside a module (the reference moves
// x/out/grad host<->device on every search call, quant_layers/linear.py:461-464).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <climits>
#include <cstdarg>
#include <cstdio>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <functional>
#include <thread>
#include <mutex>
#include <string>
#include <unordered_map>
#include <type_traits>
#include <vector>

#include "../../include/ptq4vit_hip.h"
#include "../../include/ptq4vit_hip_debug.h"
#include "p4v_kernels.h"

using namespace p4v;

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIPCHK(expr)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) return fail(P4V_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)
#define CHK(expr)              \
    do {                       \
        int r_ = (expr);       \
        if (r_ != 0) return r_; \
    } while (0)

inline long rup(long x, long m) { return (x + m - 1) / m * m; }
inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Bump allocator over the caller's workspace.  With base == nullptr it only counts (dry run):
// *_workspace_bytes() runs the same planning code as *_calibrate().
struct Arena {
    char* base;
    size_t cap, off, peak;
    size_t top;   // bytes taken from the END of the buffer: live for the whole call (candidate-plane cache), while the
                  // bump region [0, off) is rewound after every pass
    bool dry;
    Arena(void* b, size_t c) : base((char*)b), cap(c & ~(size_t)255), off(0), peak(0), top(0), dry(b == nullptr) {}
    template <typename T> T* get(size_t n) {
        off = (size_t)rup((long)off, 256);
        T* p = dry ? nullptr : (T*)(base + off);
        off += n * sizeof(T);
        if (off + top > peak) peak = off + top;
        return p;
    }
    char* get_top(size_t bytes) {
        top += (size_t)rup((long)bytes, 256);
        if (off + top > peak) peak = off + top;
        return (dry || top > cap) ? nullptr : base + cap - top;
    }
    bool ok() const { return dry || off + top <= cap; }
};

// ---- optional per-launch timing of the sweep kernels (bench.py roofline) -----------------------
// Per CALLING THREAD: a thread that enabled timing (p4v_stats_enable) gets an event pair around every sweep launch
// it enqueues and reads its own totals back with p4v_stats_get.  Nothing here is shared between threads, so calls
// on other threads / streams / devices are unaffected.
struct StatRec { hipEvent_t a, b; int kind; double macs, alg; int stage, gx, gz; double ms, bytes; };
thread_local double g_alg_bytes = 0;       // compulsory bytes of the pass being launched: its fp32 operands + raw_out / raw_grad, once
thread_local double g_alg_macs_cand = 0;   // unpadded single-plane MACs of one candidate of the pass being launched
thread_local double g_exec_frac = 1.0;     // share of a launch's candidates that a pruned pass executes (stats mode only)
thread_local bool g_stat_on = false;
thread_local std::vector<StatRec> g_stat_recs;
thread_local std::vector<StatRec> g_stat_done;   // drained records, kept until p4v_stats_reset (p4v_stats_launches)
thread_local int g_stage = 0;                   // which stage of a pruned pass is being launched: 0 full sweep, 1 A, 2 B1, 3 B2
thread_local p4v_kernel_stats g_stats = {};
thread_local long g_memo_hits = 0, g_memo_misses = 0;
// Process-wide counters of the exact candidate pruning (p4v_prune_counters): tests and bench.py assert with them that the
// three-stage passes -- not the full sweeps -- produced a result, whatever thread / stream the calibrator ran the module on.
//   [0] passes run in three stages   [1] ... whose stage B2 was empty (B1's candidates were the only survivors; host knew)
//   [2] prunable passes that ran the full sweep instead (slice too large / loose bounds / unsupported layout)
//   [3] passes not eligible at all (score tables requested, cosine, fp32 planes, pruning switched off)
std::atomic<long long> g_prune_cnt[4];
// Kernel-variant switches for A/B measurements and kernel-vs-kernel agreement tests (p4v_debug_set_variant; 0 in
// production).  One relaxed atomic word, read once per pass:
//   4   no stationary-operand sweeps (everything on k_sweep2)      8   k_sweep4 instead of k_sweep5 (one candidate per pass)
//   16  no k_sweep6 (stationary operand in LDS instead of registers)  32  k_sweep6 with 8 waves (two per SIMD)
//   64  no folding of the twin's negative plane in the activation search   128  old candidate-group heuristic
//   256 no k_sweep2g (one candidate per pass at large K)            512  no pass memoisation
//   1024 no candidate-plane cache (every pass re-packs its candidate-expanded operand)
//   2048 cosine Linear searches on k_sweep2 (swapped operands, one GEMM per V block) instead of k_sweep6
//   4096 quant_forward / folded-target GEMMs on the generic k_sweep instead of k_sweep2
//   8192 no candidate groups for the generic k_sweep
//   16384 k_sweep6 without the separate launch of the last, partial wave of workgroups
//   32768 no k_sweep7 (K >= 1024 sweeps on k_sweep2 / k_sweep2g)      65536 no k_sweep8 (single-k-tile sweeps on k_sweep2)
//   2097152 post-GELU twin of k_sweep7 on two streamed planes (not the merged one)
//   4194304 no exact candidate pruning (every candidate over every sample)   8388608 prune even where the slice's bounds are loose
//   16777216 pruned passes with several score blocks: stage B1 on the hull of the winners (no synthetic candidate)
//   33554432 the two fixed planes of a twin row operand from two k_pack launches (not k_pack_dual)
//   67108864 pruned Linear passes never try the half-size slice first
//   134217728 cross-check: every pruned pass is followed by the full sweep of the same pass, a differing selection is an error
//   1, 2: kernel debug flags (SweepParams::dbg)
//   bit 30: route every int8 sweep through the generic k_sweep
std::atomic<int> g_variant_word{0};
#define g_variant (g_variant_word.load(std::memory_order_relaxed) & 0x3fffffff)
#define g_force_v1 ((g_variant_word.load(std::memory_order_relaxed) >> 30) & 1)
// tuning overrides of the launch heuristics (p4v_debug_set_tuning; <= 0: use the cost model)
std::atomic<int> g_tune[16];
enum { TUNE_CG6 = 0, TUNE_CG2 = 1, TUNE_CG2G = 2, TUNE_CG7 = 3, TUNE_PRINT = 4, TUNE_ORDER7 = 5, TUNE_P6 = 6, TUNE_PLANE_GIB = 7, TUNE_EPI6W = 8,
       TUNE_LOOSE_PCT = 9, TUNE_SLICE_DIV = 10, TUNE_SLICE_SMALL = 11, TUNE_B1_PATH = 12, TUNE_TIER2 = 13, TUNE_TIER2_DIV = 14, TUNE_LOOSE_ROWS = 15 };   // LOOSE_ROWS: sample rows from which a module prunes on a 5 % slice share   // TIER2: 1 = no second slice tier; >= 2: minimum survivor count that triggers it   // SLICE_SMALL: rows of the slice a Linear tries first   // pruning: weight share below which a module keeps full sweeps (%); Linear slice = M / div   // EPI6W: 1 = fragment-order epilogue image also in the weight search   // P6: k_sweep6 prologue, 0.1 us; PLANE_GIB: plane budget per chunk (cache limit = half)
// TUNE_B1_PATH (key 12) doubles as the A/B switch of the round-4 / round-5 paths: 1 / 2 the bound pass on k_sweep2 / k_sweep4,
// 3 the bound pass on the sweep kernels, 5 padded 64-column planes, 6 no slice kernels, 7 cosine on the generic kernel,
// 8 read-backs by copy (no mapped host memory), 9 no per-score-block candidate ranges, 10 k_slice_b instead of k_slice_b2,
// 11 quant_fast1 instead of quant16_sat8, 12 launch geometry not planned for the host-known candidate range, >= 16 k_bound timing ablations
inline int tune(int k) { return g_tune[k].load(std::memory_order_relaxed); }

struct Group;
struct Ctx {
    hipStream_t st;
    Arena ws;
    bool dry;
    Group* grp = nullptr;     // p4v_calibrate_group: this call is member `slot` of a group whose stream operations are deferred and
    int slot = 0;             // issued together (grouped launches); nullptr: every operation goes to `st` at once
    int par = 1;              // members of the group that search in lock step: the launch heuristics plan for 256 / par CUs
};

// ---- stream operations: issued at once (one module) or deferred and grouped (p4v_calibrate_group) -------------------------------
// Every kernel launch, fill and copy of the calibration path goes through enqueue() / q_fill() / q_d2h() / q_h2d() / q_sync().
// Without a group they act on c.st immediately.  With one, the operations of a member are queued in its own FIFO and the member
// runs ahead until it needs a result on the host (q_sync: the survivor range of a pruned pass, an interval for the pass memo);
// when every member of the group waits (or has finished), the last one to arrive merges the FIFOs -- repeatedly taking the most
// common head operation over all members and issuing those launches as ONE grouped launch (k_x_g) -- synchronises the stream
// once and releases everybody.  The members' operations never depend on each other (separate modules, separate scratch), every
// member's own order is kept, and a grouped launch runs each member's body on its own parameters: bit-identical results.
struct StatInfo { int kind; double macs, alg; int stage, gx, gz; double bytes; };   // a timed sweep launch (bench.py roofline)
struct QOp;
struct KernelDesc {
    hipError_t (*one)(const QOp&, hipStream_t);
    hipError_t (*many)(const QOp* const*, int, hipStream_t);   // nullptr: the kernel has no grouped entry point
    int cap;                                                    // members per grouped launch
};
constexpr int QOP_PARAM_BYTES = 480;
struct QOp {
    enum Type : int { KERNEL, FILL, D2H, H2D, COPY2D, WAIT };
    int type = KERNEL;
    const KernelDesc* kd = nullptr;
    dim3 grid, block;
    unsigned lds = 0;
    alignas(16) char params[QOP_PARAM_BYTES];
    void* dst = nullptr; const void* src = nullptr; size_t bytes = 0; int value = 0;
    size_t dpitch = 0, spitch = 0, width = 0, height = 0;
    std::vector<char> payload;      // H2D: the host data (the caller's buffer may be gone when the operation is issued)
    bool timed = false;             // record the launch (bench.py roofline)
    bool heavy = false;             // a sweep: the issuer holds it back until no light operation is left at any member's head, so that
    StatInfo si{};                  // members that lag a few small launches behind join the same grouped launch
};
inline hipError_t lds_attr(const void* fn, unsigned lds, std::atomic<bool>& done) {
    if (lds <= 48 * 1024 || done.load(std::memory_order_acquire)) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) done.store(true, std::memory_order_release);
    return e;
}
template <typename P, void (*K1)(P), void (*KG)(GroupArgs<P>)> struct Kern {
    static_assert(sizeof(P) <= QOP_PARAM_BYTES, "kernel parameter block larger than a queued operation holds");
    static hipError_t one(const QOp& op, hipStream_t st) {
        static std::atomic<bool> attr{false};
        if (hipError_t e = lds_attr((const void*)K1, op.lds, attr); e != hipSuccess) return e;
        P p;
        std::memcpy(&p, op.params, sizeof(P));
        void* args[] = {&p};
        return hipLaunchKernel((const void*)K1, op.grid, op.block, args, op.lds, st);
    }
    static hipError_t many(const QOp* const* ops, int n, hipStream_t st) {
        static std::atomic<bool> attr{false};
        GroupArgs<P> a;
        unsigned off = 0, lds = 0;
        a.n = n;
        for (int m = 0; m < n; ++m) {
            const QOp& op = *ops[m];
            a.off[m] = off; a.gx[m] = op.grid.x; a.gy[m] = op.grid.y; a.gz[m] = op.grid.z;
            std::memcpy(&a.p[m], op.params, sizeof(P));
            off += (unsigned)rup((long)op.grid.x * op.grid.y * op.grid.z, 8);
            lds = std::max(lds, op.lds);
        }
        a.off[n] = off;
        if (hipError_t e = lds_attr((const void*)KG, lds, attr); e != hipSuccess) return e;
        void* args[] = {&a};
        return hipLaunchKernel((const void*)KG, dim3(off), ops[0]->block, args, lds, st);
    }
    static const KernelDesc* desc() { static const KernelDesc d{&one, &many, GroupArgs<P>::CAP}; return &d; }
};
template <typename P, void (*K1)(P)> struct Kern1 {      // kernels off the grouped path (generic fallbacks, API helpers)
    static_assert(sizeof(P) <= QOP_PARAM_BYTES, "kernel parameter block larger than a queued operation holds");
    static hipError_t one(const QOp& op, hipStream_t st) {
        static std::atomic<bool> attr{false};
        if (hipError_t e = lds_attr((const void*)K1, op.lds, attr); e != hipSuccess) return e;
        P p;
        std::memcpy(&p, op.params, sizeof(P));
        void* args[] = {&p};
        return hipLaunchKernel((const void*)K1, op.grid, op.block, args, op.lds, st);
    }
    static const KernelDesc* desc() { static const KernelDesc d{&one, nullptr, 1}; return &d; }
};
#define KERN(P, NAME) (Kern<P, NAME, NAME##_g>::desc())
#define KERN_T(P, NAME, ...) (Kern<P, NAME<__VA_ARGS__>, NAME##_g<__VA_ARGS__>>::desc())
#define KERN1_T(P, NAME, ...) (Kern1<P, NAME<__VA_ARGS__>>::desc())

constexpr int MIR_SLOT = 2048, MIR_SLOTS = 3;
// header of a mirror block, in ints: [0,1] / [2,3] survivor ranges of the two tiers, [4] the slice's weight share, [16..79] /
// [80..143] the per-score-block ranges of the two tiers (<= MIR_BLK blocks: the heads of an attention matmul), then the interval slots
constexpr int MIR_HDR = 160, MIR_BLK = 32, MIR_RB1 = 16, MIR_RB2 = 80;
constexpr size_t MIRROR_BYTES = sizeof(int) * MIR_HDR + sizeof(float) * MIR_SLOT * MIR_SLOTS;
inline int* mirror_alloc() {
    void* p = nullptr;
    if (hipHostMalloc(&p, MIRROR_BYTES, hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) { (void)hipGetLastError(); p = nullptr; }
    return (int*)p;
}

// The rendezvous of one p4v_calibrate_group call (see above).
struct Group {
    hipStream_t st = nullptr;
    int n = 0;
    std::mutex mu;
    std::condition_variable cv;
    std::vector<std::deque<QOp>> q;      // member FIFOs: written by the member while it runs, read by the issuer while every member is blocked
    std::vector<int> state;              // 0 running, 1 waiting in q_sync, 2 finished
    int blocked = 0;
    unsigned long gen = 0;
    int status = 0;                      // first HIP error of an issue round (every waiting member returns it)
    std::string err;
    std::vector<int*> mirrors;           // one mapped host block per member (host_mirror)
    bool stat_on = false;
    std::vector<StatRec>* stat_recs = nullptr;      // the CALLING thread's launch records: one per grouped launch
    long n_ops = 0, n_launches = 0, n_rounds = 0;
    hipEvent_t inputs_ready = nullptr;   // the members' captured tensors are complete once this event has fired (nullptr: they are)
    bool waited = false;                 // ... and the stream already waits for it
    int last_par[16] = {};               // members in the last grouped launch of each sweep family (StatInfo.kind): what the members'
                                         // launch heuristics plan for (written by the issuer while every member is blocked)
    int issue_kernels(std::vector<const QOp*>& ops);
    int issue_round();                   // (mu held, every member blocked)
    int sync(int slot);
    void finish(int slot);
};

// Process-wide counters (p4v_launch_counters): [0] kernel launches the calibration path asked for (one module at a time these ARE
// the launches), [1] kernel launches issued to the GPU, [2] issue rounds of the groups, [3] group calls.
std::atomic<long long> g_launch_cnt[4];
int enqueue_now(hipStream_t st, const QOp& op, bool timed, std::vector<StatRec>* recs) {
    g_launch_cnt[1].fetch_add(1, std::memory_order_relaxed);
    StatRec rec{};
    if (timed) {
        HIPCHK(hipEventCreate(&rec.a));
        HIPCHK(hipEventCreate(&rec.b));
        HIPCHK(hipEventRecord(rec.a, st));
    }
    HIPCHK(op.kd->one(op, st));
    if (timed) {
        HIPCHK(hipEventRecord(rec.b, st));
        rec.kind = op.si.kind; rec.macs = op.si.macs; rec.alg = op.si.alg; rec.stage = op.si.stage; rec.gx = op.si.gx; rec.gz = op.si.gz; rec.bytes = op.si.bytes;
        recs->push_back(rec);
    }
    return 0;
}
int issue_plain(hipStream_t st, const QOp& op) {
    switch (op.type) {
        case QOp::FILL: HIPCHK(hipMemsetAsync(op.dst, op.value, op.bytes, st)); break;
        case QOp::D2H: HIPCHK(hipMemcpyAsync(op.dst, op.src, op.bytes, hipMemcpyDeviceToHost, st)); break;
        case QOp::H2D: HIPCHK(hipMemcpyAsync(op.dst, op.payload.data(), op.bytes, hipMemcpyHostToDevice, st)); break;
        case QOp::COPY2D: HIPCHK(hipMemcpy2DAsync(op.dst, op.dpitch, op.src, op.spitch, op.width, op.height, hipMemcpyDeviceToDevice, st)); break;
        default: break;
    }
    return 0;
}
// one grouped launch per <= cap members of `ops` (all of one kernel entry point and block shape), in balanced chunks
int Group::issue_kernels(std::vector<const QOp*>& ops) {
    const KernelDesc* kd = ops[0]->kd;
    const int total = (int)ops.size();
    if (ops[0]->heavy && ops[0]->si.kind >= 0 && ops[0]->si.kind < 16) last_par[ops[0]->si.kind] = total;
    if (!kd->many || total == 1) {
        for (const QOp* op : ops) { CHK(enqueue_now(st, *op, op->timed && stat_recs, stat_recs)); ++n_launches; }
        return 0;
    }
    const int chunks = cdiv(total, kd->cap), per = cdiv(total, chunks);
    for (int i0 = 0; i0 < total; i0 += per) {
        const int m = std::min(per, total - i0);
        bool timed = stat_recs != nullptr;
        for (int i = 0; i < m; ++i) timed = timed && ops[i0 + i]->timed;
        StatRec rec{};
        if (timed) {
            HIPCHK(hipEventCreate(&rec.a));
            HIPCHK(hipEventCreate(&rec.b));
            HIPCHK(hipEventRecord(rec.a, st));
        }
        HIPCHK(kd->many(ops.data() + i0, m, st));
        g_launch_cnt[1].fetch_add(1, std::memory_order_relaxed);
        ++n_launches;
        if (timed) {      // ONE record per kernel launch (1:1 with a kernel trace): the members' work added up
            HIPCHK(hipEventRecord(rec.b, st));
            // (grid_x = the flat grid of the grouped launch -- every member's blocks, padded to a multiple of 8 --, as a kernel trace
            // shows it; grid_z = the number of members)
            const StatInfo& s0 = ops[i0]->si;
            rec.kind = s0.kind; rec.stage = s0.stage; rec.gz = m;
            for (int i = 0; i < m; ++i) {
                const QOp& o = *ops[i0 + i];
                rec.macs += o.si.macs; rec.alg += o.si.alg; rec.bytes += o.si.bytes;
                rec.gx += (int)rup((long)o.grid.x * o.grid.y * o.grid.z, 8);
            }
            stat_recs->push_back(rec);
        }
    }
    return 0;
}
int Group::issue_round() {
    ++n_rounds;
    g_launch_cnt[2].fetch_add(1, std::memory_order_relaxed);
    std::vector<const QOp*> bucket;
    for (;;) {
        // the most common head operation over the members (kernel entry point + block shape; plain operations go one by one);
        // light operations first: a sweep waits until every member that can still reach one has
        int best = -1, best_n = 0;
        bool best_heavy = true;
        int n_wait = 0, n_live = 0;
        for (int i = 0; i < n; ++i) {
            if (q[i].empty()) continue;
            ++n_live;
            const QOp& h = q[i].front();
            if (h.type == QOp::WAIT) { ++n_wait; continue; }                    // (last: see below)
            if (h.type != QOp::KERNEL) { best = i; best_n = 0; break; }        // fills / copies: at once, in member order
            if (h.heavy && !best_heavy) continue;
            int cnt = 0;
            for (int j = i; j < n; ++j)
                if (!q[j].empty()) { const QOp& o = q[j].front(); cnt += o.type == QOp::KERNEL && o.kd == h.kd && o.block.x == h.block.x && o.block.y == h.block.y && o.block.z == h.block.z; }
            if ((best_heavy && !h.heavy) || cnt > best_n) { best = i; best_n = cnt; best_heavy = h.heavy; }
        }
        if (best < 0 && n_wait > 0) {
            // Every member that still has work stands before the point where it needs its captured tensors: only now does the
            // stream wait for the capture -- what the members queued before (weight abs-max, candidate tables, the candidate
            // planes of the weights) ran while the capture passes were still on the GPU.
            if (!waited && inputs_ready) { HIPCHK(hipStreamWaitEvent(st, inputs_ready, 0)); waited = true; }
            for (int i = 0; i < n; ++i)
                if (!q[i].empty() && q[i].front().type == QOp::WAIT) q[i].pop_front();
            continue;
        }
        if (best < 0) break;
        const QOp& h = q[best].front();
        if (h.type != QOp::KERNEL) {
            const int r = issue_plain(st, h);
            ++n_ops;
            q[best].pop_front();
            if (r) return r;
            continue;
        }
        bucket.clear();
        std::vector<int> who;
        for (int j = best; j < n; ++j)
            if (!q[j].empty()) { const QOp& o = q[j].front(); if (o.type == QOp::KERNEL && o.kd == h.kd && o.block.x == h.block.x && o.block.y == h.block.y && o.block.z == h.block.z) { bucket.push_back(&o); who.push_back(j); } }
        n_ops += (long)bucket.size();
        const int r = issue_kernels(bucket);
        for (int j : who) q[j].pop_front();
        if (r) return r;
    }
    return 0;
}
int Group::sync(int slot) {
    std::unique_lock<std::mutex> lk(mu);
    state[slot] = 1;
    ++blocked;
    if (blocked == n) {
        int r = issue_round();
        if (!r && hipStreamSynchronize(st) != hipSuccess) r = fail(P4V_ERR_HIP, "hipStreamSynchronize failed: %s", hipGetErrorString(hipGetLastError()));
        if (r && !status) { status = r; err = g_err; }
        blocked = 0;
        for (int i = 0; i < n; ++i) { if (state[i] == 1) state[i] = 0; else if (state[i] == 2) ++blocked; }
        ++gen;
        cv.notify_all();
    } else {
        const unsigned long g = gen;
        cv.wait(lk, [&] { return gen != g; });
    }
    if (status) { g_err = err; return status; }
    return 0;
}
void Group::finish(int slot) {
    std::unique_lock<std::mutex> lk(mu);
    state[slot] = 2;
    ++blocked;
    if (blocked < n) return;
    bool waiting = false;
    for (int i = 0; i < n; ++i) waiting = waiting || state[i] == 1;
    int r = issue_round();
    if (!r && waiting && hipStreamSynchronize(st) != hipSuccess) r = fail(P4V_ERR_HIP, "hipStreamSynchronize failed: %s", hipGetErrorString(hipGetLastError()));
    if (r && !status) { status = r; err = g_err; }
    blocked = 0;
    for (int i = 0; i < n; ++i) { if (state[i] == 1) state[i] = 0; else if (state[i] == 2) ++blocked; }
    ++gen;
    cv.notify_all();
}

// a kernel launch of the calibration path; `si`: the launch is one of the timed sweeps (roofline records)
template <typename P> int enqueue(Ctx& c, const KernelDesc* kd, dim3 grid, dim3 block, size_t lds, const P& p, const StatInfo* si = nullptr) {
    if (c.dry) return 0;
    QOp op;
    op.type = QOp::KERNEL; op.kd = kd; op.grid = grid; op.block = block; op.lds = (unsigned)lds;
    static_assert(sizeof(P) <= QOP_PARAM_BYTES, "parameter block too large");
    std::memcpy(op.params, &p, sizeof(P));
    if (si) { op.heavy = true; op.timed = g_stat_on; op.si = *si; }
    g_launch_cnt[0].fetch_add(1, std::memory_order_relaxed);
    if (c.grp) { c.grp->q[c.slot].push_back(std::move(op)); return 0; }
    return enqueue_now(c.st, op, op.timed, &g_stat_recs);
}
int q_fill(Ctx& c, void* dst, int value, size_t bytes) {        // (k_fill_bytes: a kernel of this library, so that fills join the grouped launches)
    if (c.dry || bytes == 0) return 0;
    const FillBytesParams p{dst, value, (long)bytes};
    const long items = (bytes & 3) == 0 ? (long)(bytes >> 2) : (long)bytes;
    return enqueue(c, KERN(FillBytesParams, k_fill_bytes), dim3((unsigned)std::min<long>(cdiv(items, 256), 2048)), dim3(256), 0, p, nullptr);
}
int q_d2h(Ctx& c, void* host, const void* dev, size_t bytes) {       // `host` must stay valid until the next q_sync
    if (!c.grp) { HIPCHK(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, c.st)); return 0; }
    QOp op; op.type = QOp::D2H; op.dst = host; op.src = dev; op.bytes = bytes;
    c.grp->q[c.slot].push_back(std::move(op));
    return 0;
}
int q_h2d(Ctx& c, void* dev, const void* host, size_t bytes) {       // (callers synchronise before `host` goes away)
    if (!c.grp) { HIPCHK(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, c.st)); return 0; }
    QOp op; op.type = QOp::H2D; op.dst = dev; op.bytes = bytes;
    op.payload.assign((const char*)host, (const char*)host + bytes);
    c.grp->q[c.slot].push_back(std::move(op));
    return 0;
}
int q_copy2d(Ctx& c, void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height) {
    if (!c.grp) { HIPCHK(hipMemcpy2DAsync(dst, dpitch, src, spitch, width, height, hipMemcpyDeviceToDevice, c.st)); return 0; }
    QOp op; op.type = QOp::COPY2D; op.dst = dst; op.src = src; op.dpitch = dpitch; op.spitch = spitch; op.width = width; op.height = height;
    c.grp->q[c.slot].push_back(std::move(op));
    return 0;
}
// from here on the call reads its captured tensors (raw_input / raw_out / raw_grad): inside a group whose caller handed over a
// "capture done" event, the stream waits for it -- once, when every member has reached this point
int q_wait_inputs(Ctx& c) {
    if (c.dry || !c.grp || !c.grp->inputs_ready) return 0;
    QOp op; op.type = QOp::WAIT;
    c.grp->q[c.slot].push_back(std::move(op));
    return 0;
}
int q_sync(Ctx& c) {
    if (c.grp) return c.grp->sync(c.slot);
    HIPCHK(hipStreamSynchronize(c.st));
    return 0;
}

// Small device -> host results (the survivor range of a pruned pass, the slice's weight share, the intervals a pass selected)
// are written by their kernel into mapped host memory as well and read after the stream synchronisation the host does anyway:
// no copy command (a pageable hipMemcpyAsync is a blit kernel into a staging buffer between two waits: ~410 range read-backs
// and ~270 interval read-backs per ViT-B calibration).  One block per (device, stream) -- per MEMBER inside a group --, allocated
// at first use (the calibrator's streams are persistent), never freed: 32 ints ([0,1] / [2,3] survivor ranges of the two tiers,
// [4] weight share, [8..15] / [16..23] their per-score-block ranges) + MIR_SLOTS interval vectors of MIR_SLOT floats.  The null
// stream has no block (calls from different threads would share it): it takes the copy path, as p4v_debug_set_tuning(12, 8) does.
int* host_mirror(Ctx& c) {
    if (c.grp) return c.grp->mirrors[c.slot];
    if (!c.st) return nullptr;
    static std::mutex mu;
    static std::unordered_map<unsigned long long, int*> tab;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    const unsigned long long key = (unsigned long long)(uintptr_t)c.st ^ ((unsigned long long)(dev + 1) << 56);
    std::lock_guard<std::mutex> lk(mu);
    auto it = tab.find(key);
    if (it != tab.end()) return it->second;
    int* p = mirror_alloc();
    tab[key] = p;
    return p;
}
// The interval vectors of the *_impl call running on this thread that are mirrored (MirrorScope binds them, launch_select /
// k_prune_hull write through attach_mirror, read_dev reads).  `valid`: the last writer of the whole device vector was a
// mirrored selection (not the min-max initialisation, not a memo restore).
struct IvMirror { const float* dev; float* host; bool valid; int count; };
thread_local IvMirror g_mir[MIR_SLOTS] = {};
IvMirror* mirror_of(const float* dev) {
    if (!dev) return nullptr;
    for (auto& m : g_mir) if (m.dev == dev) return &m;
    return nullptr;
}
void attach_mirror(SelectParams& sl) {
    sl.iv_host = nullptr; sl.aux_host = nullptr;
    // whatever this selection writes, the mirrors of its outputs are stale until it proves otherwise
    IvMirror* mi = mirror_of(sl.interval);
    IvMirror* ma = mirror_of(sl.aux_out);
    if (mi) mi->valid = false;
    if (ma) ma->valid = false;
    if (mi && sl.nj <= MIR_SLOT && sl.out_off == 0 && (sl.nj == 1 || sl.out_js == 1)) { sl.iv_host = mi->host; mi->valid = true; mi->count = sl.nj; }
    if (ma && sl.nj <= MIR_SLOT) { sl.aux_host = ma->host; ma->valid = true; ma->count = sl.nj; }
}

void metric_epi(int metric, int* epi, int* wt_mode) {
    switch (metric) {
        case P4V_METRIC_L1_NORM: *epi = EPI_ABS; *wt_mode = 0; break;
        case P4V_METRIC_L2_NORM: *epi = EPI_SQ; *wt_mode = 0; break;
        case P4V_METRIC_LINEAR_WEIGHTED_L2: *epi = EPI_W_SQ; *wt_mode = 3; break;
        case P4V_METRIC_SQUARE_WEIGHTED_L2: *epi = EPI_SQ_W; *wt_mode = 2; break;
        case P4V_METRIC_HESSIAN: *epi = EPI_SQ_W; *wt_mode = 1; break;
        default: *epi = EPI_COS; *wt_mode = 0; break;
    }
}

// ---- launch helpers ---------------------------------------------------------------------------
int launch_absmax(Ctx& c, const float* src, const long (&st)[4], int D0, int D1, int R, int C, int nV, int nH,
                  int crb_r, int crb_c, int signed_max, unsigned* out) {
    if (c.dry) return 0;
    CHK(q_fill(c, out, 0, sizeof(unsigned) * (size_t)D1 * nV * nH));  // 0 < enc(x) for every float x
    AbsMaxParams p{src, st[0], st[1], st[2], st[3], D0, D1, R, C, nV, nH, crb_r, crb_c, 0, signed_max, out};
    p.row_tile = std::max(1, std::min(crb_r, std::max(1, 8192 / std::max(1, std::min(crb_c, C)))));
    const int tiles_per_v = cdiv(crb_r, p.row_tile);
    dim3 grid(tiles_per_v * nV * nH, D1, D0);
    return enqueue(c, KERN(AbsMaxParams, k_absmax), grid, dim3(256), 0, p);
}

int launch_interval(Ctx& c, const unsigned* enc, int n, float denom, int broadcast, float* interval) {
    if (c.dry) return 0;
    return enqueue(c, KERN(IntervalParams, k_interval_from_max), dim3(cdiv(n, 64)), dim3(64), 0, IntervalParams{enc, n, denom, broadcast, interval});
}

int launch_cands(Ctx& c, const float* mult, const float* interval, int ncand, int nblk, float* cands) {
    if (c.dry) return 0;
    return enqueue(c, KERN(CandsParams, k_make_cands), dim3(cdiv((long)ncand * nblk, 256)), dim3(256), 0, CandsParams{mult, interval, ncand, nblk, cands});
}

int launch_scale(Ctx& c, ScaleParams p) {
    if (c.dry) return 0;
    return enqueue(c, KERN(ScaleParams, k_scale_table), dim3(cdiv((long)p.C * p.nblk, 256)), dim3(256), 0, p);
}

// The rounding of v_cvt_pk_u8_f32 (k_probe_cvt, once per process) -> the bias of quant16_sat8: 128.5 where the conversion
// truncates, 128 where it rounds to nearest; 0 = not usable (the kernels keep quant_fast1).  The probe writes 8 words of a
// scratch allocation of its own (never the caller's workspace or output), under a mutex, on the caller's stream; only a probe
// that RAN and returned unexpected values latches "unusable" -- a failed launch / copy / synchronisation (a capturing stream, ...)
// leaves the state unknown: this call keeps quant_fast1 and the next one probes again.
std::atomic<int> g_cvt_state{0};      // 0 unknown, 1 truncates, 2 rounds to nearest, 3 unusable
void cvt_probe(hipStream_t stream) {
    if (g_cvt_state.load(std::memory_order_acquire) != 0) return;
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    if (g_cvt_state.load(std::memory_order_acquire) != 0) return;
    static unsigned* scratch = nullptr;
    if (!scratch && hipMalloc(&scratch, 8 * sizeof(unsigned)) != hipSuccess) { (void)hipGetLastError(); scratch = nullptr; return; }
    unsigned h[5] = {9, 9, 9, 9, 9};
    hipLaunchKernelGGL(k_probe_cvt, dim3(1), dim3(64), 0, stream, scratch);
    if (hipGetLastError() != hipSuccess || hipMemcpyAsync(h, scratch, sizeof h, hipMemcpyDeviceToHost, stream) != hipSuccess ||
        hipStreamSynchronize(stream) != hipSuccess) { (void)hipGetLastError(); return; }
    const bool sat = h[3] == 0 && h[4] == 255;
    const int st = (sat && h[0] == 0 && h[1] == 1 && h[2] == 2) ? 1 : (sat && h[0] == 1 && h[1] == 2 && (h[2] == 2 || h[2] == 3)) ? 2 : 3;
    if (tune(TUNE_PRINT) > 0) fprintf(stderr, "[p4v] v_cvt_pk_u8_f32(0.7, 1.5, 2.5, -3, 300) = %u %u %u %u %u -> mode %d\n", h[0], h[1], h[2], h[3], h[4], st);
    g_cvt_state.store(st, std::memory_order_release);
}
float cvt_bias(Ctx& c) {
    if (!c.dry && !c.grp) cvt_probe(c.st);         // (a group probes once, on the calling thread, before its members start)
    const int st = g_cvt_state.load(std::memory_order_acquire);
    return st == 1 ? 128.5f : st == 2 ? 128.0f : 0.0f;
}
template <typename T> int launch_pack(Ctx& c, const PackParams& p_) {
    if (c.dry) return 0;
    PackParams p = p_;
    // the full symmetric 8-bit grid: quant16_sat8 once the conversion has been probed (the *_impl entry points do that first);
    // tuning 12 = 11 keeps quant_fast1 (A/B)
    p.qbias = (sizeof(T) == 1 && p.mode == PACK_SYM && p.lo == -128 && p.hi == 127 && tune(TUNE_B1_PATH) != 11) ? cvt_bias(c) : 0.0f;
    const long total = (long)p.Z * p.Rp * (p.Kp / 16);
    if (total >= (1L << 31)) return fail(P4V_ERR_UNSUPPORTED, "operand plane too large for k_pack (%ld 16-element runs)", total);
    // (a pruned launch lets most candidate groups exit at once: fewer, longer-running workgroups)
    const int blocks = (int)std::min<long>(cdiv(total, 256), p.crange ? 256L * 12 : 256L * 64);
    if (p.mode == PACK_TWIN_I8) {
        if (sizeof(T) != 1 || p.C != 1 || p.c_inner != 0 || !p.scales || p.conv || p.zdiv > 0)
            return fail(P4V_ERR_UNSUPPORTED, "merged twin plane: one fixed row-major int8 plane only");
        return enqueue(c, KERN(PackParams, k_pack_twin), dim3(blocks), dim3(256), 0, p);
    }
    if (tune(TUNE_PRINT) > 1) fprintf(stderr, "[p4v] k_pack stage %d: Z %d Rp %d Kp %d C %d crange %d done %d layout %d blocks %d\n", g_stage, p.Z, p.Rp, p.Kp, p.C, p.crange != nullptr, p.done != nullptr, p.c_inner, blocks);
    return enqueue(c, KERN_T(PackParams, k_pack, T), dim3(blocks, cdiv(p.C, PACK_CG)), dim3(256), 0, p);
}

// (inside a group the lock-step members share the chip: each plans for its share of the workgroup slots)
// `kind`: the sweep family (StatInfo.kind) -- what the group's last launch of that family held is the better estimate of how many
// members are in lock step than the number of same-shaped members
inline int lockstep(const Ctx& c, int kind) {
    if (!c.grp) return 1;
    const int seen = (kind >= 0 && kind < 16) ? c.grp->last_par[kind] : 0;
    return std::max(1, seen > 0 ? seen : c.par);
}
inline int cu_slots(const Ctx& c, int slots, int kind) { return std::max(8, slots / lockstep(c, kind)); }
// epilogue dispatch of a kernel family: E = the metric's epilogue as a template argument
#define P4V_EPI4(epi, X)                                   \
    switch (epi) {                                         \
        case EPI_SQ_W: { constexpr int E = EPI_SQ_W; X; }  \
        case EPI_SQ: { constexpr int E = EPI_SQ; X; }      \
        case EPI_ABS: { constexpr int E = EPI_ABS; X; }    \
        default: { constexpr int E = EPI_W_SQ; X; }        \
    }

template <typename T, bool TWIN> int launch_sweep_epi(Ctx& c, const SweepParams& p, int epi, int cgroups, const StatInfo* si) {
    const size_t lds = 2 * (TWIN ? 3 : 2) * SW_TILE_BYTES;
    dim3 grid(p.mtiles * p.ntiles, p.Z, cgroups), block(512);
    switch (epi) {
        case EPI_SQ_W: return enqueue(c, KERN_T(SweepParams, k_sweep, T, TWIN, EPI_SQ_W), grid, block, lds, p, si);
        case EPI_SQ: return enqueue(c, KERN_T(SweepParams, k_sweep, T, TWIN, EPI_SQ), grid, block, lds, p, si);
        case EPI_ABS: return enqueue(c, KERN_T(SweepParams, k_sweep, T, TWIN, EPI_ABS), grid, block, lds, p, si);
        case EPI_W_SQ: return enqueue(c, KERN_T(SweepParams, k_sweep, T, TWIN, EPI_W_SQ), grid, block, lds, p, si);
        case EPI_STORE: return enqueue(c, KERN1_T(SweepParams, k_sweep, T, TWIN, EPI_STORE), grid, block, lds, p, si);
        case EPI_FWD: return enqueue(c, KERN1_T(SweepParams, k_sweep, T, TWIN, EPI_FWD), grid, block, lds, p, si);
        default: return enqueue(c, KERN1_T(SweepParams, k_sweep, T, TWIN, EPI_COS), grid, block, lds, p, si);
    }
}

// k_sweep2g: large K, column operand expanded, row operand invariant (weight search): two candidates per pass
bool sweep2g_ok(const SweepParams& p) {
    return p.b_cs != 0 && p.a_cs == 0 && p.ktiles >= 16 && (p.c1 - p.c0) >= 2 && p.o_bs == 0 && p.o_nbs == 0 &&
           !(g_variant & 256);
}

template <bool TWIN> int launch_sweep2g_epi(Ctx& c, const SweepParams& p, int epi, int cgroups, const StatInfo* si) {
    const int per = 2 * cdiv(p.c1 - p.c0, 2 * cgroups);
    const size_t lds = (size_t)SW2_NS * (TWIN ? 4 : 3) * SW2_TILE + (size_t)per * 8 * sizeof(float) * (TWIN ? 3 : 2);
    dim3 grid(p.mtiles * p.ntiles, p.Z, cgroups), block(512);
    P4V_EPI4(epi, return enqueue(c, KERN_T(SweepParams, k_sweep2g, TWIN, E), grid, block, lds, p, si))
}

template <bool TWIN> int launch_sweep2_epi(Ctx& c, const SweepParams& p, int epi, int cgroups, const StatInfo* si) {
    const int per = cdiv(p.c1 - p.c0, cgroups);
    const size_t lds = (size_t)SW2_NS * (TWIN ? 3 : 2) * SW2_TILE + (size_t)per * 8 * sizeof(float) * (TWIN ? 3 : 2);
    dim3 grid(p.mtiles * p.ntiles, p.Z, cgroups), block(512);
    switch (epi) {
        case EPI_SQ_W: return enqueue(c, KERN_T(SweepParams, k_sweep2, TWIN, EPI_SQ_W), grid, block, lds, p, si);
        case EPI_SQ: return enqueue(c, KERN_T(SweepParams, k_sweep2, TWIN, EPI_SQ), grid, block, lds, p, si);
        case EPI_ABS: return enqueue(c, KERN_T(SweepParams, k_sweep2, TWIN, EPI_ABS), grid, block, lds, p, si);
        case EPI_FWD: return enqueue(c, KERN1_T(SweepParams, k_sweep2, TWIN, EPI_FWD), grid, block, lds, p, si);
        case EPI_STORE:     // (the twin instance has no register to spare for the group prologue: single launches)
            if constexpr (TWIN) return enqueue(c, KERN1_T(SweepParams, k_sweep2, TWIN, EPI_STORE), grid, block, lds, p, si);
            else return enqueue(c, KERN_T(SweepParams, k_sweep2, TWIN, EPI_STORE), grid, block, lds, p, si);
        case EPI_COS: return enqueue(c, KERN_T(SweepParams, k_sweep2, TWIN, EPI_COS), grid, block, lds, p, si);
        default: return enqueue(c, KERN_T(SweepParams, k_sweep2, TWIN, EPI_W_SQ), grid, block, lds, p, si);
    }
}

#ifdef P4V_TRACE
// tuning builds only (-DP4V_TRACE): per-workgroup timestamps of the stationary sweeps, appended to $P4V_TRACE_FILE
unsigned long long* g_trace = nullptr;
int trace_attach(Sweep3Params& p) {
    if (!g_trace) HIPCHK(hipMalloc(&g_trace, sizeof(unsigned long long) * 16 * 65536));
    p.trace = g_trace;
    return 0;
}
int trace_dump(Ctx& c, dim3 grid, int ktiles) {
    if (!getenv("P4V_TRACE_FILE")) return 0;
    const size_t n = (size_t)grid.x * grid.z * 16;
    HIPCHK(hipStreamSynchronize(c.st));
    std::vector<unsigned long long> h(n);
    HIPCHK(hipMemcpy(h.data(), g_trace, n * 8, hipMemcpyDeviceToHost));
    FILE* f = fopen(getenv("P4V_TRACE_FILE"), "ab");
    unsigned long long hdr[4] = {0xABCDull, grid.x, grid.z, (unsigned long long)ktiles};
    fwrite(hdr, 8, 4, f); fwrite(h.data(), 8, n, f); fclose(f);
    return 0;
}
#endif

// the record of a timed sweep launch (bench.py roofline): every launch is recorded, also a stage whose device-side candidate
// range is empty -- the records are the production launches, 1:1 with a kernel trace
inline StatInfo stat_info(int kind, double macs, double alg, int gx, int gz, double bytes) {
    return StatInfo{kind, macs * g_exec_frac, alg * g_exec_frac, g_stage, gx, gz, bytes};
}

int launch_sweep4(Ctx& c, const Sweep3Params& p, int epi, int cgroups, bool pair) {
    if (c.dry) return 0;
    const int per = pair ? 2 * cdiv(p.c1 - p.c0, 2 * cgroups) : cdiv(p.c1 - p.c0, cgroups);
    const size_t lds = (size_t)p.ktiles * SW2_TILE + (size_t)(pair ? SW5_NP * 2 : SW4_NS) * SW2_TILE + (size_t)per * 8 * sizeof(float) * 2 + 256;
    dim3 grid(p.stiles * p.ttiles, 1, cgroups), block(512);
#ifdef P4V_TRACE
    CHK(trace_attach(const_cast<Sweep3Params&>(p)));
#endif
    const StatInfo si = stat_info(5, (double)p.stiles * 128 * (double)p.ttiles * 128 * (double)p.ldk * (p.c1 - p.c0), g_alg_macs_cand * (p.c1 - p.c0),
                                  (int)grid.x, (int)grid.z, g_alg_bytes);
    const StatInfo* sp = &si;
    if (pair) { P4V_EPI4(epi, CHK(enqueue(c, KERN_T(Sweep3Params, k_sweep5, E), grid, block, lds, p, sp)); break) }
    else { P4V_EPI4(epi, CHK(enqueue(c, KERN_T(Sweep3Params, k_sweep4, E), grid, block, lds, p, sp)); break) }
#ifdef P4V_TRACE
    CHK(trace_dump(c, grid, p.ktiles));
#endif
    return 0;
}

template <int KT, int RB>
int launch_sweep6_kt(Ctx& c, const Sweep3Params& p, int epi, dim3 grid, size_t lds, const StatInfo* si) {
    if constexpr (RB == 2) {       // cosine (plain = activation search, transposed = weight search): one wave per SIMD only
        if (epi == EPI_COS) return enqueue(c, KERN_T(Sweep3Params, k_sweep6, EPI_COS, KT, 2), grid, dim3(256), lds, p, si);
        if (epi == EPI_COS_T) return enqueue(c, KERN_T(Sweep3Params, k_sweep6, EPI_COS_T, KT, 2), grid, dim3(256), lds, p, si);
    }
    if (epi == EPI_COS || epi == EPI_COS_T) return fail(P4V_ERR_UNSUPPORTED, "k_sweep6: the cosine epilogue has no 8-wave instance");
    P4V_EPI4(epi, return enqueue(c, KERN_T(Sweep3Params, k_sweep6, E, KT, RB), grid, dim3(512 / RB), lds, p, si))
}

// k_sweep6 (stationary operand in registers): K = 192 / 256 / 384 / 512 / 768 bytes -- the Linear layers of
// ViT/DeiT-T/S/B and of Swin stages 2-3
bool sweep6_supported(int ktiles) { return ktiles == 3 || ktiles == 4 || ktiles == 6 || ktiles == 8 || ktiles == 12; }

int launch_sweep6_part(Ctx& c, const Sweep3Params& p, int epi, int cgroups);

// One workgroup per CU (512 registers per wave): `tiles` equal workgroups run in ceil(tiles / 256) waves and the last
// one is mostly empty (ViT-B qkv: 900 tiles = 3.5 waves).  The tiles of the last, partial wave are launched separately
// with their candidates split over q groups, so that it takes a fraction of a full wave's time; every group pays
// the workgroup prologue (stationary operand + raw_out / raw_grad tile) again.  Cost model in microseconds.
// (Inside a group the other members' tiles fill the last wave: no split.)
int launch_sweep6(Ctx& c, const Sweep3Params& p, int epi, int cgroups, int nc_model = 0) {
    if (c.dry) return 0;
    // (p.ntile > 0: only the tiles [p.tile0, p.tile0 + p.ntile) -- the open score blocks of a pruned pass; nc_model: the number of
    // candidates the device-side range leaves, when the host knows it)
    const int tiles = p.ntile > 0 ? p.ntile : p.stiles * p.ttiles, base = p.ntile > 0 ? p.tile0 : 0;
    const int nc = nc_model > 0 ? nc_model : p.c1 - p.c0;
    const int full = tiles / 256 * 256, rem = tiles - full;
    const double P = tune(TUNE_P6) > 0 ? 0.1 * tune(TUNE_P6) : 20.0, t_c = 0.196 * p.ktiles;        // prologue, one candidate of one tile
    auto waves = [](long wgs) { return (double)((wgs + 255) / 256); };
    int q_best = 0;
    if (rem > 0 && full > 0 && !(g_variant & 16384) && lockstep(c, 2) <= 1) {
        double best = waves((long)tiles * cgroups) * (P + cdiv(nc, cgroups) * t_c) * 0.97;   // the uniform plan
        for (int q = 1; q <= std::min(nc, 12); ++q) {
            const double t = waves(full) * (P + nc * t_c) + waves((long)rem * q) * (P + cdiv(nc, q) * t_c);
            if (t < best) { best = t; q_best = q; }
        }
    }
    if (tune(TUNE_PRINT) > 0) fprintf(stderr, "[p4v] sweep6 tiles %d: full %d rem %d -> q %d (uniform cg %d)\n", tiles, full, rem, q_best, cgroups);
    if (q_best > 0) {
        Sweep3Params a = p, b = p;
        a.tile0 = base; a.ntile = full;
        b.tile0 = base + full; b.ntile = rem;
        CHK(launch_sweep6_part(c, a, epi, 1));
        CHK(launch_sweep6_part(c, b, epi, q_best));
    } else {
        Sweep3Params a = p;
        a.tile0 = base; a.ntile = p.ntile > 0 ? tiles : 0;
        CHK(launch_sweep6_part(c, a, epi, cgroups));
    }
    return 0;
}

int launch_sweep6_part(Ctx& c, const Sweep3Params& p, int epi, int cgroups) {
    const int per = cdiv(p.c1 - p.c0, cgroups);
    // 32-row blocks per wave: 2 = 4 waves, one per SIMD (default: bound by its own VALU + MFMA issue, LDS-light);
    // 1 = 8 waves, two per SIMD (A/B variant 32: hides the VALU work but doubles the fragment reads -> LDS-bound;
    // both measure 3.33 ms per fc1 search round on MI355X)
    const int rb = ((g_variant & 32) && epi != EPI_COS && epi != EPI_COS_T) ? 1 : 2;
    const int nw = 8 / rb;
    const size_t lds = (size_t)3 * p.ktiles * 4096 + (size_t)(per + 1) * 2 * nw * sizeof(float) + (size_t)per * nw * sizeof(float) + 64 * sizeof(float) + 256;
    dim3 grid(p.ntile > 0 ? p.ntile : p.stiles * p.ttiles, 1, cgroups);
#ifdef P4V_TRACE
    CHK(trace_attach(const_cast<Sweep3Params&>(p)));
#endif
    // one record per kernel launch; a split sweep books its work (and the pass's bytes) in proportion to the tiles of each part
    const double share = (double)grid.x / ((double)p.stiles * p.ttiles);
    const StatInfo si = stat_info(2, share * (double)p.stiles * 256 * (double)p.ttiles * 64 * (double)p.ldk * (p.c1 - p.c0),
                                  share * g_alg_macs_cand * (p.c1 - p.c0), (int)grid.x, (int)grid.z, share * g_alg_bytes);
    const StatInfo* sp = &si;
    int r;
#define P4V_KT(K) (rb == 2 ? launch_sweep6_kt<K, 2>(c, p, epi, grid, lds, sp) : launch_sweep6_kt<K, 1>(c, p, epi, grid, lds, sp))
    switch (p.ktiles) {
        case 12: r = P4V_KT(12); break;
        case 8: r = P4V_KT(8); break;
        case 6: r = P4V_KT(6); break;
        case 4: r = P4V_KT(4); break;
        default: r = P4V_KT(3); break;
    }
#undef P4V_KT
    if (r) return r;
#ifdef P4V_TRACE
    CHK(trace_dump(c, grid, p.ktiles));
#endif
    return 0;
}

// k_sweep9: single-k-tile sweeps on 16 x 16 blocks (part layout [C][Z][halves * 8], set up by run_pass)
int sweep9_halves(const SweepParams& p, bool twin, int epi) {
    if (p.ktiles != 1 || twin || (p.a_cs == 0) == (p.b_cs == 0) || epi == EPI_STORE || epi == EPI_FWD || epi == EPI_COS) return 0;
    if (p.sb_mode == 1 && p.s_cs > 1) return 0;
    if (p.bias_axis != 0 || (g_variant & 524288)) return 0;
    // one ring stage holds the whole streamed operand of a candidate: 256 rows x 64 B (SW9_STAGE); beyond that k_sweep8
    if ((p.a_cs == 0 ? p.N : p.M) > 256) return 0;
    const long nb = (long)cdiv(p.M, 16) * cdiv(p.N, 16);
    const int halves = (int)cdiv(nb, (long)SW9_NW * SW9_NB);
    if (halves > 32 / SW9_NW) return 0;
    // worth it where the 128 x 128 tiles of k_sweep8 carry padding: Swin windows (144 tokens: 3.2 x the 16-granular area, q.k^T
    // search 9.2 -> 6.0 ms per module; 49 tokens: 4 x) and, by 5 %, the 197 tokens of ViT / DeiT (1.5 x: 433 -> 410 us per pass).
    // Variant 1048576 forces it for A/B runs, 524288 disables it.
    const double waste = (double)rup(p.M, 128) * rup(p.N, 128) / ((double)rup(p.M, 16) * rup(p.N, 16));
    return (waste >= 1.4 || (g_variant & 1048576)) ? halves : 0;
}
template <bool ROWS_FIXED> int launch_sweep9_epi(Ctx& c, const SweepParams& p, int epi, int cgroups, const StatInfo* si) {
    const int per = cdiv(p.c1 - p.c0, cgroups);
    const size_t lds = (size_t)SW9_NS * SW9_STAGE + (size_t)per * SW9_NW * sizeof(float) * 2;
    dim3 grid(p.halves, p.Z, cgroups), block(SW9_NW * 64);
    P4V_EPI4(epi, return enqueue(c, KERN_T(SweepParams, k_sweep9, ROWS_FIXED, E), grid, block, lds, p, si))
}

// k_sweep8: single k-tile (K <= 64) int8 sweep with the fixed operand's fragments in registers: q.k^T of every ViT / Swin
bool sweep8_ok(const SweepParams& p, bool twin, int epi) {
    return p.ktiles == 1 && !twin && (p.a_cs == 0) != (p.b_cs == 0) && epi != EPI_STORE && epi != EPI_FWD && epi != EPI_COS &&
           !(g_variant & 65536);
}

template <bool ROWS_FIXED, bool SKIP> int launch_sweep8_epi_s(Ctx& c, const SweepParams& p, int epi, int cgroups, const StatInfo* si) {
    const int per = cdiv(p.c1 - p.c0, cgroups);
    const size_t lds = (size_t)SW8_NS * SW2_TILE + (size_t)per * 8 * sizeof(float) * 2;
    dim3 grid(p.mtiles * p.ntiles, p.Z, cgroups), block(512);
    P4V_EPI4(epi, return enqueue(c, KERN_T(SweepParams, k_sweep8, ROWS_FIXED, E, SKIP), grid, block, lds, p, si))
}
// The padding skip of k_sweep2 (wave parts without a valid element do no MFMA / epilogue work) is available here only as an A/B
// switch (variant 262144): measured slower at 197 tokens (4 of 32 parts padding: 459 vs 428 us) AND at the 144 tokens of a Swin
// window (17 of 32 parts: 10.0 vs 9.3 ms per module) -- a single-k-tile candidate is paced by its ring step (DMA landing +
// barrier), not by the MFMAs and the epilogue it would skip.
template <bool ROWS_FIXED> int launch_sweep8_epi(Ctx& c, const SweepParams& p, int epi, int cgroups, const StatInfo* si) {
    const bool skip = (g_variant & 262144) != 0;
    return skip ? launch_sweep8_epi_s<ROWS_FIXED, true>(c, p, epi, cgroups, si) : launch_sweep8_epi_s<ROWS_FIXED, false>(c, p, epi, cgroups, si);
}

// k_sweep7: large-K int8 sweep (both operands streaming, 256 x 256 workgroup tile)
template <int TWIN> int launch_sweep7_epi(Ctx& c, const Sweep7Params& p, int epi, dim3 grid, size_t lds, const StatInfo* si) {
    if constexpr (TWIN == 0) { if (epi == EPI_COS) return enqueue(c, KERN_T(Sweep7Params, k_sweep7, 0, EPI_COS), grid, dim3(512), lds, p, si); }
    if (epi == EPI_COS) return fail(P4V_ERR_UNSUPPORTED, "k_sweep7: the cosine epilogue has no twin instance");
    P4V_EPI4(epi, return enqueue(c, KERN_T(Sweep7Params, k_sweep7, TWIN, E), grid, dim3(512), lds, p, si))
}

int launch_sweep7(Ctx& c, const Sweep7Params& p, int twin, int epi, int cgroups) {   // twin: 0 plain, 1 two planes, 2 merged plane
    if (c.dry) return 0;
    const int nc = p.c1 - p.c0;
    // the per-candidate tables live behind the ring: at most 160 candidates per workgroup
    const int per_max = (int)((160 * 1024 - SW7_NS * SW7_STAGE - 256) / 192);
    cgroups = std::max(cgroups, cdiv(nc, per_max));
    const int per = cdiv(nc, cgroups);
    const size_t lds = (size_t)SW7_NS * SW7_STAGE + (size_t)per * 192 + 256;
    Sweep7Params q = p;
    q.cgroups = cgroups;
    dim3 grid(p.rtiles * p.ctiles * cgroups, 1, 1);
    // (twin: 128 samples x 2 planes)
    const StatInfo si = stat_info(twin ? 4 : 3, (double)p.rtiles * 256 * (double)p.ctiles * 256 * (double)p.ldk * nc, g_alg_macs_cand * nc,
                                  (int)grid.x, (int)grid.z, g_alg_bytes);
    const StatInfo* sp = &si;
    return twin == 2 ? launch_sweep7_epi<2>(c, q, epi, grid, lds, sp) : twin ? launch_sweep7_epi<1>(c, q, epi, grid, lds, sp) : launch_sweep7_epi<0>(c, q, epi, grid, lds, sp);
}

int launch_sweep(Ctx& c, const SweepParams& p, bool i8, bool twin, int epi, bool fast = false, int cgroups = 1) {
    if (c.dry) return 0;
    // kernel family of the record (p4v_launch_record.kind): 12 k_bound, 6 k_sweep9, 7 k_sweep8, 8 k_sweep2g, 9 k_sweep2, 0 / 1 generic int8 / fp32
    const int kind = (fast && p.bound) ? 12 : (fast && p.halves > 0) ? 6 : (fast && sweep8_ok(p, twin, epi)) ? 7 :
                     (fast && epi != EPI_COS && sweep2g_ok(p)) ? 8 : fast ? 9 : i8 ? 0 : 1;
    const double kelems = (double)p.ldk / (i8 ? 1 : 4);
    const StatInfo si = stat_info(kind, (double)p.mtiles * SW_BM * (double)p.ntiles * SW_BN * kelems * p.Z * (p.c1 - p.c0) * (twin ? 2 : 1),
                                  g_alg_macs_cand * (p.c1 - p.c0),
                                  (fast && p.bound) ? p.mtiles * p.ntiles * 2 : (fast && p.halves > 0) ? p.halves : p.mtiles * p.ntiles, cgroups, g_alg_bytes);
    const StatInfo* sp = &si;
    if (fast && p.bound) {
        const dim3 grid(p.mtiles * p.ntiles * 2), block(256);      // 128 x 64 workgroup tiles
        P4V_EPI4(epi, return enqueue(c, KERN_T(SweepParams, k_bound, E), grid, block, 0, p, sp))
    }
    if (fast && p.halves > 0) return p.a_cs == 0 ? launch_sweep9_epi<true>(c, p, epi, cgroups, sp) : launch_sweep9_epi<false>(c, p, epi, cgroups, sp);
    if (fast && sweep8_ok(p, twin, epi)) return p.a_cs == 0 ? launch_sweep8_epi<true>(c, p, epi, cgroups, sp) : launch_sweep8_epi<false>(c, p, epi, cgroups, sp);
    if (fast && epi != EPI_COS && sweep2g_ok(p)) return twin ? launch_sweep2g_epi<true>(c, p, epi, cgroups, sp) : launch_sweep2g_epi<false>(c, p, epi, cgroups, sp);
    if (fast) return twin ? launch_sweep2_epi<true>(c, p, epi, cgroups, sp) : launch_sweep2_epi<false>(c, p, epi, cgroups, sp);
    if (i8) return twin ? launch_sweep_epi<int8_t, true>(c, p, epi, cgroups, sp) : launch_sweep_epi<int8_t, false>(c, p, epi, cgroups, sp);
    return twin ? launch_sweep_epi<float, true>(c, p, epi, cgroups, sp) : launch_sweep_epi<float, false>(c, p, epi, cgroups, sp);
}

int launch_finish(Ctx& c, const FinishParams& p) {
    return enqueue(c, KERN(FinishParams, k_finish), dim3(p.C, p.nj), dim3(256), 0, p);
}

int launch_finish_cos(Ctx& c, const FinishCosParams& p) {
    const int gy = p.j_mode == 3 ? cdiv(p.S, 256) : p.nj;
    return enqueue(c, KERN(FinishCosParams, k_finish_cos), dim3(p.C, gy), dim3(p.j_mode == 3 ? 256 : 1024), 0, p);
}

int launch_select(Ctx& c, const SelectParams& p_) {
    if (c.dry) return 0;
    SelectParams p = p_;
    attach_mirror(p);
    return enqueue(c, KERN(SelectParams, k_select), dim3(p.nj), dim3(128), 0, p);
}

// ---- one search pass -------------------------------------------------------------------------------
// A "pass" evaluates eq_n candidates of ONE operand against a fixed counterpart and selects the best
// candidate per score block.  The GEMM is D[rows][cols] = rowop . colop^T; in the plain orientation rows
// are samples (activations / matmul A) and columns are output features (weights / matmul B); the cosine
// metric runs swapped so that the feature axis it reduces over lies on the MFMA rows.
// The candidate-expanded plane of one search (weights x 100 scales, ...) depends on the operand and on the candidate
// table only -- both fixed for the whole calibration_step2 (the table is built once from the INITIAL interval,
// reference linear.py:544-545) -- so rounds 2..R find it already packed.  One object per (module, searched operand),
// owned by the *_impl call; the buffer lives at the top of the workspace.
struct PlaneCache {
    char* buf = nullptr;
    bool assigned = false, valid = false;
    unsigned char* done = nullptr;    // pruned passes: device flags, one per candidate already in `buf`
};
// The sample slice of the pruned passes (run_pass_pruned, stage A): which samples carry the metric weight depends on raw_grad /
// raw_out only, so the ranking, the gathered rows of raw_out / raw_grad and of the row operand are built by the first pruned pass
// of a module and reused by the others (both searches, all rounds); a buffer is gathered again only when its source changed
// (the folded target of the twin activation search is rebuilt per pass).
struct SliceCache {
    bool assigned = false;
    int k = 0;                        // rows per segment the buffers are sized for
    int k_eff = 0;                    // rows per segment in use (a Linear whose 256 heaviest samples hold the weight uses those)
    int* idx = nullptr; float* mass = nullptr; float* mass_u = nullptr;
    float* Os = nullptr; float* Gs = nullptr; float* Rs = nullptr; float* Cs = nullptr;
    const void* idx_src = nullptr; int idx_wt = -1;      // what the ranking was computed from
    const void* o_src = nullptr; const void* g_src = nullptr; const void* r_src = nullptr; const void* c_src = nullptr;
    float* frac = nullptr;            // device: share of the metric weight the slice holds
    PlaneCache aplane;                // stage A's candidate-expanded plane of the SLICED operand (the search whose row operand is
                                      // expanded): candidates and slice rows are fixed for the call, the later rounds find it packed
    bool loose = false;               // that share is too small for the stages to pay (Swin: no class token): full sweeps
    float frac_host = 1.0f;           // that share as the host read it (0 < f < 1 once measured): decides about the second tier
};
// The same for the epilogue operands of k_sweep6 in fragment order (k_prep_epi6): raw_out, raw_grad and the bias are fixed
// for the whole call, so the tile image of one search orientation is built by its first pass and read by the later rounds.
typedef PlaneCache EpiCache;

struct Operand {
    PackParams pk;        // src/strides/sizes/scales/mode filled by the caller (dst, C, Rp, Kp set by run_pass)
    bool expanded;        // true: one plane per candidate
    bool present;
};

struct Pass {
    bool i8, twin;
    int epi, wt_mode;
    Operand row, row2, col;   // row2 = twin second plane (always on the row side)
    int Z;                    // batched GEMMs (matmul batch*heads, V blocks of the swapped cosine sweep)
    long row_zs_shared, col_zs_shared;  // 1 = operand shared across z (stride 0)
    int Mrows, Ncols, K;      // valid sizes of the GEMM
    int eq_n;
    // scales
    ScaleParams s1, s2;       // s.S filled by run_pass; nblk = s_cs
    bool use_s1;
    int sb_mode, sb_div, s_cs;
    // epilogue operands
    const float* bias; int bias_axis; long bias_zs;
    const float* O; const float* G;
    long o_zs, o_bs, o_ms, o_nbs, o_ns; int o_inner, o_ninner;
    // finish
    int j_mode, j_div, nj; double norm;
    int cos_ZB, cos_ZV, cos_j_mode, cos_j_div;   // cosine finish geometry
    // select
    const float* cands; int cand_cs, cand_js, cand_off;
    float* interval; int out_js, out_off;
    float* aux_out; float aux_div;
    float* scores_out; int scores_out_ld;
    int32_t* best_out;
    float* store_out;         // EPI_STORE pass: one "candidate", writes raw_out - bias - scale*acc, no finish/select
    bool cos6;                // cosine of a plain Linear layer on k_sweep6 (rows = samples, cols = features, Z = 1): set by linear_impl
    bool cos7;                // ... on k_sweep7 (K >= 1024)
    bool twin_disjoint;       // twin whose two ranges never overlap (post-GELU): k_sweep7 may stream them as one merged plane
    // exact candidate pruning (run_pass_pruned): device-side candidate range, scores kept for the next stage, no selection
    const int* crange;
    const int* crange_blk;    // ... and per score block [2 * nj] (k_prune_hull's rblk): honoured where the sweep kernel's tiles lie inside
                              // one score block (run_pass decides; otherwise every block sweeps `crange`, a superset)
    int host_lo, host_hi;     // what the host read of `crange` (host_hi > host_lo: known) -- the launch geometry is planned for the
    const int* host_rblk;     // candidates that will run -- and of `crange_blk` (nj <= MIR_BLK; nullptr: unknown): closed blocks are not launched
    float* scores_keep;
    bool no_select;
    bool prunable;            // set by the *_impl callers for passes whose score is minus a sum of non-negative terms
    bool prunable_f32;        // ... and which may be pruned although their operands are fp32 planes (conv with a_bit >= 32)
    float* S1_pre; float* S2_pre; bool s_ready;   // scale tables shared by the stages of a pruned pass (same table, same scales)
    SliceCache* scache;       // optional: the module's sample slice, shared by its pruned passes
    SliceCache* scache2;      // optional (Linear): the module's second, larger slice (two-tier pruning, run_pass_pruned)
    bool bound_kernel;        // stage B1 of a pruned pass: ONE candidate over all samples -> k_bound where the layout allows it
    bool host_sync_ok;        // the caller synchronises the stream after the pass anyway (pass memo): the pruned pass may read
                              // the 8-byte survivor range back and skip the launches of an empty stage B2
    PlaneCache* cache;        // optional: keeps the candidate-expanded plane across the rounds of one call
    EpiCache* ecache;         // optional: keeps k_sweep6's fragment-order epilogue operands across the rounds of one call
    bool pack_only;           // plan the pass and pack its candidate-expanded operand into `cache`, nothing else (linear_impl: the
                              // candidate planes of the WEIGHTS depend on no captured tensor -- packed while the capture still runs)
};

static const long PLANE_BUDGET_DEFAULT = 6L << 30;  // bytes of candidate-expanded plane kept resident per chunk
#define PLANE_BUDGET (tune(TUNE_PLANE_GIB) > 0 ? ((long)tune(TUNE_PLANE_GIB) << 30) : PLANE_BUDGET_DEFAULT)
#define PLANE_CACHE_MAX (PLANE_BUDGET / 2)      // largest candidate-expanded plane kept across the rounds of one call

// How many candidate groups (gridDim.z) to split a sweep into.  Splitting raises the workgroup count (fills the
// 256 CUs / evens out the last round) but every workgroup pays its prologue (raw_out/raw_grad tile, stationary
// operand) again.  Cost model in microseconds, constants measured on MI355X (profiles/): minimise
// rounds * (prologue + candidates_per_group * ktiles * tile_time).
int choose_cgroups(long wgs, int ncand, int ktiles, int slots, double prologue_us, double tile_us, int cg_max = 25) {
    int best = 1;
    double best_t = 1e30;
    for (int cg = 1; cg <= std::min(ncand, cg_max); ++cg) {
        const long rounds = (wgs * cg + slots - 1) / slots;
        const double t = (double)rounds * (prologue_us + (double)cdiv(ncand, cg) * ktiles * tile_us);
        if (t < best_t * 0.999) { best_t = t; best = cg; }
    }
    return best;
}

// the two fixed planes of a twin row operand can come from one launch: same source view, one scale each, plain modes
bool dual_pack_ok(const PackParams& a, const PackParams& b) {
    auto plain = [](const PackParams& p) {
        return (p.mode == PACK_SYM || p.mode == PACK_SOS_HI || p.mode == PACK_SOS_LO) && p.blk_mode == 0 && !p.conv && !p.crange;
    };
    return plain(a) && plain(b) && a.src == b.src && a.R == b.R && a.K == b.K && a.s_r == b.s_r && a.s_k == b.s_k &&
           a.s_z == b.s_z && a.s_z2 == b.s_z2 && a.zdiv == b.zdiv;
}

int run_pass(Ctx& c, Pass& ps) {
    const int esz = ps.i8 ? 1 : 4;
    g_alg_macs_cand = (double)ps.Mrows * ps.Ncols * ps.K * ps.Z;
    // every cached tensor read once per search pass -- both operands in fp32 as captured, raw_out and the metric weight
    g_alg_bytes = 4.0 * ((double)ps.Mrows * ps.K * (ps.row_zs_shared ? 1 : ps.Z) + (double)ps.Ncols * ps.K * (ps.col_zs_shared ? 1 : ps.Z)) +
                  (ps.G ? 8.0 : 4.0) * (double)ps.Mrows * ps.Ncols * ps.Z;
    const int Kp = (int)rup(ps.K, 64 / esz);          // 64-byte k-tiles
    // stationary-operand sweep (k_sweep4): Linear layers whose invariant operand tile (128 x K int8) fits in LDS
    const bool blocks64 = (ps.s_cs == 1 || ps.sb_div % 64 == 0) &&
                          (ps.j_mode == 0 || (ps.j_mode == 1 && (ps.nj == 1 || ps.j_div % 64 == 0)));
    // stage B1 (one candidate over all samples): k_bound -- rows = samples, columns = features of a plain [M][N] layer, whole
    // 32-column groups inside one score block (the partial-sum table of the fast sweeps)
    const bool bound = ps.bound_kernel && ps.i8 && !ps.twin && ps.Z == 1 && !ps.store_out && ps.epi != EPI_COS && !g_force_v1 &&
                       ps.o_bs == 0 && ps.o_nbs == 0 && ps.bias_axis == 0 && (ps.sb_mode == 0 || ps.sb_mode == 1) &&
                       (ps.j_mode == 0 || (ps.j_mode == 1 && (ps.nj == 1 || ps.j_div % 32 == 0))) &&
                       (ps.sb_mode != 1 || ps.s_cs == 1 || ps.sb_div % 32 == 0) && (ps.eq_n == 1 || ps.crange);
    const bool b1_generic = bound || (g_stage == 2 && tune(TUNE_B1_PATH) == 1);    // (tuning 12=1: experiment, the bound pass on k_sweep2)
    const bool stat_ok = !b1_generic && !ps.store_out && ps.i8 && !ps.twin && (ps.epi != EPI_COS || ps.cos6) && !g_force_v1 && !(g_variant & 4) && ps.Z == 1 &&
                         ps.sb_mode == 1 && blocks64 && (ps.row.expanded != ps.col.expanded) &&
                         rup(ps.K, 64) <= 768 && ps.o_bs == 0 && ps.o_nbs == 0;
    const bool b1_lds = g_stage == 2 && tune(TUNE_B1_PATH) == 2;        // experiment: the bound pass on k_sweep4 (stationary operand in LDS)
    const bool regs6 = stat_ok && sweep6_supported(Kp / SW_BKB) && !(g_variant & 16) && !b1_lds;   // k_sweep6: stationary operand in registers
    if (ps.cos6 && !regs6) return fail(P4V_ERR_UNSUPPORTED, "cosine pass planned for k_sweep6 does not qualify for it");
    const bool pairs = stat_ok && !regs6 && !(g_variant & 8) && !b1_lds;                // k_sweep5: two candidates per pass
    // per-score-block candidate ranges: the weight search on k_sweep6 (a streaming 64-row tile lies in one scale block = one score
    // block: blocks64); tuning 12 = 9 switches them off (A/B)
    const int* rblk = (ps.crange && ps.crange_blk && regs6 && !ps.row.expanded && ps.j_mode == 1 && ps.nj == ps.s_cs &&
                       ps.j_div == ps.sb_div && tune(TUNE_B1_PATH) != 9) ? ps.crange_blk : nullptr;
    // ... and the head-wise searches of the attention matmuls (score block = z % heads: a workgroup works on one z)
    const bool rblk_z_ok = ps.crange && ps.crange_blk && !stat_ok && ps.j_mode == 2 && ps.nj == ps.j_div && ps.Z > 1 && ps.epi != EPI_COS &&
                           !ps.store_out && !bound && rup(ps.K, 64) < 1024 && tune(TUNE_B1_PATH) != 9;
    const int* rblk_z = rblk_z_ok ? ps.crange_blk : nullptr;
    const int PADR = SW_BM;
    // k_sweep7 (large K): rows = samples, columns = output features of a plain [M][N] layer; features contiguous in
    // raw_out / raw_grad and a multiple of 32 (dwordx4 epilogue loads, whole 32-feature blocks), every 32-feature block
    // inside one scale / score block, exactly one operand candidate-expanded, the twin's second plane not expanded
    const bool big7 = !b1_generic && !stat_ok && !ps.store_out && ps.i8 && (ps.epi != EPI_COS || ps.cos7) && !g_force_v1 && !(g_variant & 32768) && ps.Z == 1 &&
                      rup(ps.K, 64) >= 1024 && rup(ps.K, 64) % 256 == 0 && (long)ps.Mrows * ps.o_ms * 4 < (1L << 32) && ps.sb_mode == 1 && (ps.s_cs == 1 || ps.sb_div % 32 == 0) &&
                      (ps.j_mode == 0 || (ps.j_mode == 1 && (ps.nj == 1 || ps.j_div % 32 == 0))) &&
                      ps.row.expanded != ps.col.expanded && !(ps.twin && (ps.row.expanded || ps.row2.expanded)) &&
                      ps.o_bs == 0 && ps.o_nbs == 0 && ps.o_ns == 1 && ps.Ncols % 32 == 0 && ps.o_ms % 4 == 0 &&
                      (c.dry || ((((unsigned long long)ps.O) | ((unsigned long long)(ps.G ? ps.G : ps.O))) & 15) == 0) && ps.bias_axis == 0;
    // post-GELU twin on k_sweep7: ONE merged int8 plane k_pos + k_neg (disjoint supports), split in registers (variant
    // 2097152 keeps the two-plane kernel for A/B runs)
    const bool merged7 = big7 && ps.twin && ps.twin_disjoint && !(g_variant & 2097152);
    if (ps.cos7 && (!big7 || ps.twin)) return fail(P4V_ERR_UNSUPPORTED, "cosine pass planned for k_sweep7 does not qualify for it");
    // k_sweep6 tiles the stationary operand (the one that is NOT candidate-expanded) in 256-row slabs
    const int Mp = (int)rup(ps.Mrows, big7 ? (ps.twin ? 128 : 256) : (regs6 && !ps.row.expanded) ? 256 : PADR);
    const int Np = (int)rup(ps.Ncols, big7 ? 256 : (regs6 && !ps.col.expanded) ? 256 : PADR);
    // rows of the column operand's plane: the tile-padded count, except for the 64-column operands of the batched k_sweep2 passes
    // (attn.v: B = V, N = head_dim = 64 -- 100 candidate planes of 384 x 128 x 256 B = 1.26 GB per ViT-B module half of which was
    // padding: 630 us of k_pack and the stream of k_sweep2's stage A): k_sweep2 re-reads rows 0..63 for the tile's upper half
    const bool fast_pre = !stat_ok && !(ps.store_out && (g_variant & 4096)) && ps.i8 && ps.epi != EPI_COS && !(g_force_v1) &&
                          (ps.sb_mode != 1 || ps.s_cs == 1 || ps.sb_div % 32 == 0) &&
                          (ps.j_mode == 0 || ps.j_mode == 2 || (ps.j_mode == 1 && (ps.nj == 1 || ps.j_div % 32 == 0)));
    const bool b64 = fast_pre && !big7 && !bound && ps.Z > 1 && !ps.col_zs_shared && ps.Ncols <= 64 && Kp / SW_BKB >= 2 && Kp / SW_BKB <= 15 &&
                     !ps.store_out && tune(TUNE_B1_PATH) != 5;
    const int NpB = b64 ? 64 : Np;
    const long row_plane = (long)ps.Z * Mp * Kp * esz, col_plane = (long)ps.Z * NpB * Kp * esz;
    const long row_plane1 = ps.row_zs_shared ? (long)Mp * Kp * esz : row_plane;
    const long col_plane1 = ps.col_zs_shared ? (long)NpB * Kp * esz : col_plane;
    const long exp_plane = ps.row.expanded ? row_plane1 : col_plane1;
    int chunk = (int)std::max<long>(1, std::min<long>(ps.eq_n, PLANE_BUDGET / std::max<long>(1, exp_plane)));

    const size_t mark = c.ws.off;
    const size_t slack = stat_ok ? 4096 : 0;   // k_sweep4's ring keeps issuing a few tiles past the last candidate
    if (pairs && chunk > 1) chunk &= ~1;                  // chunks start on a candidate pair
    const int chunk_al = pairs ? ((chunk + 1) & ~1) : chunk;   // an odd count is padded to a whole pair
    // (planes above PLANE_CACHE_MAX are re-packed every pass: with three search streams and two searched operands per
    // module the cache would otherwise add up to 6 x PLANE_BUDGET of workspace on the 128-image configurations)
    PlaneCache* pc = (ps.cache && chunk >= ps.eq_n && !ps.store_out && ps.row.expanded != ps.col.expanded &&
                      !(ps.twin && ps.row2.expanded) && exp_plane * (long)ps.eq_n <= PLANE_CACHE_MAX &&
                      !(g_variant & 1024)) ? ps.cache : nullptr;
    if (pc && !pc->assigned) {
        pc->buf = c.ws.get_top((size_t)exp_plane * chunk_al + slack);
        pc->done = reinterpret_cast<unsigned char*>(c.ws.get_top((size_t)rup(ps.eq_n, 256)));
        pc->assigned = true; pc->valid = false;
        if (!c.dry && pc->done) CHK(q_fill(c, pc->done, 0, (size_t)ps.eq_n));
    }
    char* rowbuf = (pc && ps.row.expanded) ? pc->buf : c.ws.get<char>((size_t)row_plane1 * (ps.row.expanded ? chunk_al : 1) + slack);
    char* row2buf = ps.twin ? c.ws.get<char>((size_t)row_plane1 * (ps.row2.expanded ? chunk : 1)) : nullptr;
    char* colbuf = (pc && ps.col.expanded) ? pc->buf : c.ws.get<char>((size_t)col_plane1 * (ps.col.expanded ? chunk_al : 1) + slack);
    const int MT = Mp / 64;
    const bool cosm = ps.epi == EPI_COS;
    // fast int8 sweep (k_sweep2): needs every 32-column group inside one scale block and one score block
    const bool fast = !stat_ok && !(ps.store_out && (g_variant & 4096)) && ps.i8 && !cosm && !(g_force_v1) &&
                      (ps.sb_mode != 1 || ps.s_cs == 1 || ps.sb_div % 32 == 0) &&
                      (ps.j_mode == 0 || ps.j_mode == 2 || (ps.j_mode == 1 && (ps.nj == 1 || ps.j_div % 32 == 0)));
    // cosine on k_sweep2 (same stream and ring; three sums per sample and wave in k_sweep's table layout, k_finish_cos unchanged)
    const bool fast_cos = cosm && !stat_ok && !big7 && ps.i8 && !ps.store_out && !g_force_v1 && tune(TUNE_B1_PATH) != 7 &&
                          (ps.sb_mode != 1 || ps.s_cs == 1 || ps.sb_div % 32 == 0);
    // k_sweep4 table: [slabs of 64 stationary rows][groups of 32 streaming rows]
    const bool a_search = ps.row.expanded;          // stationary = weights (col operand), streaming = activations
    const int s3_gw = 32;                           // streaming rows per wave (column group width of the table)
    const int s3_slabs = (a_search ? Np : Mp) / 64, s3_groups = (a_search ? Mp : Np) / s3_gw;
    const int NpP = stat_ok ? s3_groups : fast ? Np / 32 : Np;          // columns of the partial-sum table
    const int MT7 = big7 ? (Mp / (ps.twin ? 128 : 256)) * 4 : 0;        // k_sweep7: one row per (sample tile, wave column)
    // cosine on k_sweep6: k_finish_cos's table [64-feature slab][padded sample][3]; the samples are the streaming rows of the
    // activation search (tiles of 64) and the stationary rows of the weight search (slabs of 256)
    // ... on k_sweep7: slabs of 128 features (a wave's rows), the samples are the tile-padded rows
    const int cos_slab = ps.cos7 ? 128 : 64;
    const int cos6_Sp = ps.cos7 ? Mp : ps.cos6 ? (a_search ? (int)cdiv(ps.Mrows, 64) * 64 : Mp) : 0;
    const int cos6_slabs = ps.cos7 ? Np / 128 : ps.cos6 ? (a_search ? Np / 64 : (int)cdiv(ps.Ncols, 64)) : 0;
    const long p_zs = (ps.cos6 || ps.cos7) ? (long)cos6_slabs * cos6_Sp * 3 : stat_ok ? (long)s3_slabs * s3_groups : big7 ? (long)MT7 * NpP : (long)MT * NpP * (cosm ? 3 : 1);
    const long p_cs = p_zs * ps.Z;
    float* part = c.ws.get<float>((size_t)p_cs * ps.eq_n);
    float* S1 = !ps.use_s1 ? nullptr : ps.S1_pre ? ps.S1_pre : c.ws.get<float>((size_t)ps.eq_n * ps.s_cs);
    float* S2 = !(ps.use_s1 && ps.twin) ? nullptr : ps.S2_pre ? ps.S2_pre : c.ws.get<float>((size_t)ps.eq_n * ps.s_cs);
    float* scores = ps.scores_keep ? ps.scores_keep : c.ws.get<float>((size_t)ps.eq_n * std::max(1, ps.nj));
    float* zero_bias = (stat_ok && !ps.bias) ? c.ws.get<float>((size_t)std::max(Mp, Np)) : nullptr;
    float* epi7 = big7 ? c.ws.get<float>((size_t)Mp * Np * 2) : nullptr;   // k_sweep7: epilogue operands in fragment order
    // k_sweep6: epilogue operands in fragment order, one image per 256 x 64 tile (8 bytes per output element)
    const int s6_stiles = (a_search ? Np : Mp) / 256, s6_ttiles = (int)cdiv(a_search ? ps.Mrows : ps.Ncols, 64);
    // (only where the in-place gather is uncoalesced: the activation search, whose tile is transposed; in the weight search the
    // lanes of a load already read consecutive features)
    const bool epi6_on = regs6 && (a_search || ps.cos6 || tune(TUNE_EPI6W) == 1);
    const size_t epi6_bytes = epi6_on ? (size_t)s6_stiles * s6_ttiles * (256 * 64 * 8) : 0;
    EpiCache* ec = (epi6_on && ps.ecache && (long)epi6_bytes <= PLANE_CACHE_MAX && !(g_variant & 1024)) ? ps.ecache : nullptr;
    if (ec && !ec->assigned) {
        ec->buf = c.ws.get_top(epi6_bytes);
        ec->assigned = true; ec->valid = false;
    }
    float* epi6 = !epi6_on ? nullptr : ec ? reinterpret_cast<float*>(ec->buf) : c.ws.get<float>(epi6_bytes / 4);
    if (!c.ws.ok()) return fail(P4V_ERR_WORKSPACE, "workspace too small: need >= %zu bytes", c.ws.off);
    auto pack = [&](Operand& op, char* buf, int Rp, bool shared, int c0, int nc) -> int {
        PackParams pk = op.pk;
        pk.Rp = Rp; pk.Kp = Kp; pk.dst = buf;
        pk.Z = shared ? 1 : ps.Z;
        pk.C = op.expanded ? nc : 1;
        pk.c_inner = (stat_ok && op.expanded) ? (pairs ? 2 : 1) : 0;   // k_sweep4 / k_sweep5 stream [row][candidate][K]
        if (regs6 && !op.expanded) pk.c_inner = 3;                     // k_sweep6: stationary operand in MFMA-fragment order
        if (bound) pk.c_inner = 3;                                     // k_bound: both operands (one candidate each) in fragment order
        if (op.expanded && pk.scales) pk.scales += (long)c0 * pk.sc_cs;
        if (op.expanded && ps.crange) {       // pruned pass: only the candidate groups in range, and not the ones already kept
            pk.crange = ps.crange; pk.c_base = c0;
            pk.done = (pc && pc->done) ? pc->done : nullptr;
        }
        CHK(ps.i8 ? launch_pack<int8_t>(c, pk) : launch_pack<float>(c, pk));
        return 0;      // (the finish of this pass flags the groups as packed: FinishParams.mark_done)
    };
    if (ps.pack_only) {
        // every candidate of the expanded operand into the module's plane cache, in the layout this pass's sweep streams; the
        // pass itself (and every later one of the module) then finds the plane packed
        if (pc && !pc->valid && ps.row.expanded != ps.col.expanded && !ps.crange) {
            if (ps.row.expanded) CHK(pack(ps.row, rowbuf, Mp, ps.row_zs_shared, 0, ps.eq_n));
            else CHK(pack(ps.col, colbuf, NpB, ps.col_zs_shared, 0, ps.eq_n));
            pc->valid = true;
            if (!c.dry && pc->done) CHK(q_fill(c, pc->done, 1, (size_t)ps.eq_n));
        }
        c.ws.off = mark;
        return 0;
    }
    if (epi6_on && !c.dry && !(ec && ec->valid)) {
        PrepEpi6Params pe{};
        pe.O = ps.O; pe.Wt = ps.G ? ps.G : ps.O; pe.bias = ps.bias ? ps.bias : zero_bias;
        pe.o_ss = a_search ? ps.o_ns : ps.o_ms; pe.o_ts = a_search ? ps.o_ms : ps.o_ns;
        pe.SR = a_search ? ps.Ncols : ps.Mrows; pe.TR = a_search ? ps.Mrows : ps.Ncols;
        pe.bias_on_t = a_search ? 0 : 1; pe.wt_mode = ps.cos6 ? 4 : ps.wt_mode;
        pe.stiles = s6_stiles; pe.ttiles = s6_ttiles; pe.E = epi6; pe.transposed = (ps.cos6 && !a_search) ? 1 : 0;
        if (zero_bias) CHK(q_fill(c, zero_bias, 0, sizeof(float) * (size_t)std::max(Mp, Np)));
        const long chunks = (long)(epi6_bytes / 16);
        CHK(enqueue(c, KERN(PrepEpi6Params, k_prep_epi6), dim3((unsigned)std::min<long>(cdiv(chunks, 256), 256L * 32)), dim3(256), 0, pe));
    }
    if (ec) ec->valid = true;
    if (big7 && !c.dry) {
        PrepEpiParams pe{ps.O, ps.G ? ps.G : ps.O, ps.bias, ps.o_ms, ps.Mrows, ps.Ncols, ps.cos7 ? 4 : ps.wt_mode,
                         Np / 256, Mp / (ps.twin ? 128 : 256), ps.twin ? 1 : 0, epi7};
        const long chunks = (long)Mp * Np * 2 / 4;
        CHK(enqueue(c, KERN(PrepEpiParams, k_prep_epi), dim3((unsigned)std::min<long>(cdiv(chunks, 256), 256L * 16)), dim3(256), 0, pe));
    }

    if (zero_bias && !c.dry) CHK(q_fill(c, zero_bias, 0, sizeof(float) * (size_t)std::max(Mp, Np)));
    if (ps.use_s1 && !ps.s_ready) {
        ps.s1.S = S1; ps.s1.C = ps.eq_n; ps.s1.nblk = ps.s_cs;
        CHK(launch_scale(c, ps.s1));
        if (ps.twin) { ps.s2.S = S2; ps.s2.C = ps.eq_n; ps.s2.nblk = ps.s_cs; CHK(launch_scale(c, ps.s2)); }
    }
    // fixed planes once
    if (merged7) {
        Operand both = ps.row;              // positive range: scale + upper clamp; negative range: fixed scale + lower clamp
        both.pk.mode = PACK_TWIN_I8;
        both.pk.lo = ps.row2.pk.lo; both.pk.neg_scale = ps.row2.pk.neg_scale;
        CHK(pack(both, rowbuf, Mp, ps.row_zs_shared, 0, 1));
    } else if (ps.twin && ps.i8 && !ps.row.expanded && !ps.row2.expanded && dual_pack_ok(ps.row.pk, ps.row2.pk) && !(g_variant & 33554432)) {
        // both planes of the twin row operand from one read of the source (k_pack_dual)
        PackParams p1 = ps.row.pk, p2 = ps.row2.pk;
        p1.Rp = p2.Rp = Mp; p1.Kp = p2.Kp = Kp; p1.Z = p2.Z = ps.row_zs_shared ? 1 : ps.Z; p1.C = p2.C = 1;
        p1.dst = rowbuf; p2.dst = row2buf;
        if (!c.dry) {
            const long total = (long)p1.Z * Mp * (Kp / 16);
            if (total >= (1L << 31)) return fail(P4V_ERR_UNSUPPORTED, "operand plane too large for k_pack_dual (%ld 16-element runs)", total);
            CHK(enqueue(c, KERN(PackDualParams, k_pack_dual), dim3((unsigned)std::min<long>(cdiv(total, 256), 256L * 64)), dim3(256), 0, PackDualParams{p1, p2}));
        }
    } else {
        if (!ps.row.expanded) CHK(pack(ps.row, rowbuf, Mp, ps.row_zs_shared, 0, 1));
        if (ps.twin && !ps.row2.expanded) CHK(pack(ps.row2, row2buf, Mp, ps.row_zs_shared, 0, 1));
    }
    if (!ps.col.expanded) CHK(pack(ps.col, colbuf, NpB, ps.col_zs_shared, 0, 1));

    int nine_halves = 0;
    for (int c0 = 0; c0 < ps.eq_n; c0 += chunk) {
        const int nc = std::min(chunk, ps.eq_n - c0);
        const bool packed = pc && pc->valid;   // (a cached plane is never chunked: one iteration)
        if (ps.row.expanded && !packed) CHK(pack(ps.row, rowbuf, Mp, ps.row_zs_shared, c0, nc));
        if (ps.twin && ps.row2.expanded) CHK(pack(ps.row2, row2buf, Mp, ps.row_zs_shared, c0, nc));
        if (ps.col.expanded && !packed) CHK(pack(ps.col, colbuf, NpB, ps.col_zs_shared, c0, nc));
        if (pc && !ps.crange && !packed) {   // every candidate is in the buffer now (a pruned pass packs a range and keeps flags)
            pc->valid = true;
            if (!c.dry && pc->done) CHK(q_fill(c, pc->done, 1, (size_t)ps.eq_n));
        }
        if (g_stat_on && ps.crange && !c.dry) {     // roofline step only: how many of this launch's candidates run
            // From what the host knows WITHOUT another round trip wherever it does -- the timed calibration must make the same
            // launches, in the same issue rounds, as an untimed one (inside a group an extra synchronisation regroups the members):
            // the survivor range the pruned pass read back anyway (stages A2 / B2 under the pass memo), or "one candidate" (stage B1
            // of a single score block: the range holds the slice winner).  Only a caller without either reads the range back here.
            int h[2] = {0, 0};
            const bool host_knows = ps.host_hi > ps.host_lo;
            const bool one_cand = !host_knows && g_stage == 2 && ps.nj == 1;
            if (host_knows) { h[0] = ps.host_lo; h[1] = ps.host_hi; }
            else if (one_cand) { h[0] = c0; h[1] = c0 + 1; }
            else { CHK(q_d2h(c, h, ps.crange, sizeof h)); CHK(q_sync(c)); }
            const int lo = std::max(h[0], c0), hi = std::min(h[1], c0 + nc);
            g_exec_frac = (double)std::max(0, hi - lo) / (double)nc;
            if ((rblk || rblk_z) && ps.nj <= 64 && !one_cand) {               // equal-sized score blocks, each on its own range
                int hb[128];
                if (host_knows && ps.host_rblk && ps.nj <= MIR_BLK) std::copy(ps.host_rblk, ps.host_rblk + 2 * ps.nj, hb);
                else { CHK(q_d2h(c, hb, rblk ? rblk : rblk_z, sizeof(int) * 2 * ps.nj)); CHK(q_sync(c)); }
                double sum = 0;
                for (int j = 0; j < ps.nj; ++j) sum += std::max(0, std::min(std::min(hb[2 * j + 1], h[1]), c0 + nc) - std::max(std::max(hb[2 * j], h[0]), c0));
                g_exec_frac = sum / ((double)nc * ps.nj);
            }
            if (tune(TUNE_PRINT) > 0) fprintf(stderr, "[p4v] pruned stage: candidates [%d, %d) of %d  (M %d N %d K %d Z %d nj %d)\n", h[0], h[1], ps.eq_n, ps.Mrows, ps.Ncols, ps.K, ps.Z, ps.nj);
        } else g_exec_frac = 1.0;
        if (stat_ok) {
            Sweep3Params q{};
            q.S = a_search ? colbuf : rowbuf; q.s_zs = 0;
            q.T = a_search ? rowbuf : colbuf; q.t_cs = 0; q.t_zs = 0;
            q.t_rs = (long)(pairs ? ((nc + 1) & ~1) : nc) * Kp;   // [row][candidate][K] layout written by k_pack (c_inner)
            q.ldk = Kp; q.ktiles = Kp / SW_BKB;
            q.S1 = S1; q.s_cs = ps.s_cs; q.sb_on_t = a_search ? 0 : 1; q.sb_div = std::max(1, ps.sb_div);
            q.bias = ps.bias ? ps.bias : zero_bias; q.bias_on_t = a_search ? 0 : 1;
            q.O = ps.O; q.Wt = ps.G ? ps.G : ps.O; q.wt_mode = ps.wt_mode;
            q.o_ss = a_search ? ps.o_ns : ps.o_ms; q.o_ts = a_search ? ps.o_ms : ps.o_ns;
            q.SR = a_search ? ps.Ncols : ps.Mrows; q.TR = a_search ? ps.Mrows : ps.Ncols;
            q.c0 = c0; q.c1 = c0 + nc; q.crange = ps.crange;
            q.part = part; q.p_cs = p_cs; q.NG = s3_groups;
            q.stiles = (a_search ? Np : Mp) / 128; q.ttiles = (a_search ? Mp : Np) / 128;
            q.dbg = g_variant & 3;
            if (regs6) {
                // streaming tiles of 64 rows: only those holding valid rows (the plane is padded to 128)
                q.stiles = s6_stiles; q.ttiles = s6_ttiles; q.E = epi6; q.crange_blk = rblk;
                const int epi6k = ps.cos6 ? (a_search ? EPI_COS : EPI_COS_T) : ps.epi;
                if (ps.cos6) q.NG = cos6_Sp;
                // what the host knows of the device-side range: the launch geometry is planned for the candidates that will run
                const bool known = ps.crange && ps.host_hi > ps.host_lo && tune(TUNE_B1_PATH) != 12;
                const int nc_known = known ? std::max(1, std::min(ps.host_hi, c0 + nc) - std::max(ps.host_lo, c0)) : nc;
                const double P6 = tune(TUNE_P6) > 0 ? 0.125 * tune(TUNE_P6) : 25.0;
                if (rblk && known && ps.host_rblk && ps.nj <= 4) {
                    // per-score-block ranges known to the host: one launch per run of OPEN blocks (a closed block -- its only survivor
                    // is its stage-A winner -- has nothing to sweep: the q block of a ViT qkv layer); score block j = streaming
                    // tiles [j, j + 1) * sb_div / 64
                    const int tpb = ps.sb_div / 64;
                    for (int j = 0; j < ps.nj;) {
                        if (ps.host_rblk[2 * j + 1] <= ps.host_rblk[2 * j]) { ++j; continue; }
                        int j1 = j, lo = INT_MAX, hi = 0;
                        double fsum = 0;
                        for (; j1 < ps.nj && ps.host_rblk[2 * j1 + 1] > ps.host_rblk[2 * j1]; ++j1) {
                            const int l = std::max(ps.host_rblk[2 * j1], c0), h = std::min(ps.host_rblk[2 * j1 + 1], c0 + nc);
                            lo = std::min(lo, l); hi = std::max(hi, h);
                            fsum += std::max(0, h - l);
                        }
                        Sweep3Params qq = q;
                        const int tt0 = j * tpb, tt1 = std::min(q.ttiles, j1 * tpb);
                        qq.tile0 = tt0 * q.stiles; qq.ntile = (tt1 - tt0) * q.stiles;
                        const int ncr = std::max(1, hi - lo);
                        g_exec_frac = fsum / ((double)(j1 - j) * nc);
                        int cg6 = choose_cgroups((long)qq.ntile, ncr, q.ktiles, cu_slots(c, 256, 2), P6, 0.14);
                        if (tune(TUNE_CG6) > 0) cg6 = std::max(1, std::min(ncr, tune(TUNE_CG6)));
                        if (tune(TUNE_PRINT) > 0) fprintf(stderr, "[p4v] sweep6 open blocks [%d, %d): tiles %d x %d ktiles %d cand %d -> cgroups %d\n", j, j1, q.stiles, tt1 - tt0, q.ktiles, ncr, cg6);
                        if (hi > lo && qq.ntile > 0) CHK(launch_sweep6(c, qq, epi6k, cg6, ncr));
                        j = j1;
                    }
                    continue;
                }
                int cg6 = choose_cgroups((long)q.stiles * q.ttiles, nc_known, q.ktiles, cu_slots(c, 256, 2), P6, 0.14);
                if (tune(TUNE_CG6) > 0) cg6 = std::max(1, std::min(nc, tune(TUNE_CG6)));
                if (tune(TUNE_PRINT) > 0) fprintf(stderr, "[p4v] sweep6 tiles %d x %d ktiles %d cand %d (%d known) -> cgroups %d\n", q.stiles, q.ttiles, q.ktiles, nc, nc_known, cg6);
                CHK(launch_sweep6(c, q, epi6k, cg6, known ? nc_known : 0));
                continue;
            }
            const long wgs = (long)q.stiles * q.ttiles;
            const int cgroups = choose_cgroups(wgs, nc, q.ktiles, cu_slots(c, 256, 5), 30.0, 0.15);
            CHK(launch_sweep4(c, q, ps.epi, cgroups, pairs));
            continue;
        }
        if (big7) {
            Sweep7Params q{};
            q.r_cs = ps.col.expanded ? col_plane1 : 0;
            q.R = colbuf - (long)c0 * q.r_cs;
            q.c_cs = ps.row.expanded ? row_plane1 : 0;
            q.Cp = rowbuf - (long)c0 * q.c_cs;
            q.C2 = (ps.twin && !merged7) ? row2buf : nullptr;
            q.ldk = Kp; q.ktiles = Kp / SW_BKB;
            q.S1 = S1; q.S2 = S2; q.s_cs = ps.s_cs; q.sb_div = ps.s_cs > 1 ? std::max(1, ps.sb_div) : (1 << 30);
            q.E = epi7;
            q.c0 = c0; q.c1 = c0 + nc; q.crange = ps.crange;
            q.part = part; q.p_cs = p_cs; q.NG = ps.cos7 ? cos6_Sp : NpP;
            q.rtiles = Np / 256; q.ctiles = Mp / (ps.twin ? 128 : 256);
            // one workgroup per CU; per k-tile ~0.62 us (16 MFMAs per wave, two waves per SIMD), ~3 k-tiles' worth of
            // epilogue per candidate, a prologue of a few us (scale tables, first tiles)
            // (up to one candidate per workgroup: stage A of a pruned pass is 3 tiles x 100 candidates -- with the 25 groups of the
            // other sweeps 75 workgroups on 256 CUs, 140-160 us per launch)
            int cg7 = choose_cgroups((long)q.rtiles * q.ctiles, nc, q.ktiles + 3, cu_slots(c, 256, ps.twin ? 4 : 3), 6.0, 0.62, 100);
            if (tune(TUNE_CG7) > 0) cg7 = std::max(1, std::min(nc, tune(TUNE_CG7)));
            q.order = tune(TUNE_ORDER7) > 0 ? tune(TUNE_ORDER7) - 1 : 1;
            if (tune(TUNE_PRINT) > 0) fprintf(stderr, "[p4v] sweep7 tiles %d x %d ktiles %d cand %d twin %d -> cgroups %d\n", q.rtiles, q.ctiles, q.ktiles, nc, (int)ps.twin, cg7);
            CHK(launch_sweep7(c, q, merged7 ? 2 : ps.twin ? 1 : 0, ps.epi, cg7));
            continue;
        }
        SweepParams sp{};
        // plane pointers are biased so that the kernel can index them with the absolute candidate id
        sp.a_cs = ps.row.expanded ? row_plane1 : 0;
        sp.A = rowbuf - (long)c0 * sp.a_cs;
        sp.a_zs = ps.row_zs_shared ? 0 : (long)Mp * Kp * esz;
        if (ps.twin) {
            sp.a2_cs = ps.row2.expanded ? row_plane1 : 0;
            sp.A2 = row2buf - (long)c0 * sp.a2_cs;
            sp.a2_zs = sp.a_zs;
        }
        sp.b_cs = ps.col.expanded ? col_plane1 : 0;
        sp.B = colbuf - (long)c0 * sp.b_cs;
        sp.b_zs = ps.col_zs_shared ? 0 : (long)NpB * Kp * esz;
        sp.b_rows = b64 ? 64 : 0;
        sp.ldk = Kp * esz; sp.ktiles = sp.ldk / SW_BKB;
        sp.S1 = S1; sp.S2 = S2; sp.s_cs = ps.s_cs; sp.sb_mode = ps.sb_mode; sp.sb_div = std::max(1, ps.sb_div);
        sp.bias = ps.bias; sp.bias_axis = ps.bias_axis; sp.bias_zs = ps.bias_zs;
        sp.O = ps.O; sp.Wt = ps.G ? ps.G : ps.O; sp.wt_mode = ps.wt_mode;
        sp.o_zs = ps.o_zs; sp.o_bs = ps.o_bs; sp.o_ms = ps.o_ms; sp.o_nbs = ps.o_nbs; sp.o_ns = ps.o_ns;
        sp.o_inner = ps.o_inner > 0 ? ps.o_inner : INT_MAX;
        sp.o_ninner = ps.o_ninner > 0 ? ps.o_ninner : INT_MAX;
        sp.M = ps.Mrows; sp.N = ps.Ncols; sp.Z = ps.Z; sp.c0 = c0; sp.c1 = c0 + nc; sp.crange = ps.crange;
        sp.crange_blk = rblk_z; sp.cb_div = std::max(1, ps.j_div);
        sp.part = part; sp.p_cs = p_cs; sp.p_zs = p_zs; sp.Np = NpP;
        sp.mtiles = Mp / SW_BM; sp.ntiles = Np / SW_BN;
        sp.dbg = g_variant & 3;
        sp.store = ps.store_out;
        int cgroups = 1;
        if (!fast && !fast_cos && !(g_variant & 8192)) {
            // generic sweep: 2 workgroups per CU; per k-tile step ~2.6 us with fp32 operands (8 x mfma_f32_32x32x2 per
            // 32x32 block), ~1.6 us on the int8 grid (measured on the patch-embedding search)
            const long wgs = (long)sp.mtiles * sp.ntiles * ps.Z;
            cgroups = choose_cgroups(wgs, nc, sp.ktiles, cu_slots(c, 512, ps.i8 ? 0 : 1), 20.0, ps.i8 ? 1.6 : 2.6);
        }
        if (fast || fast_cos) {
            const long wgs = (long)sp.mtiles * sp.ntiles * ps.Z;
            cgroups = (g_variant & 128) ? (int)std::max<long>(1, std::min<long>(std::min(nc, 10), (2048 + wgs - 1) / wgs))
                                        : choose_cgroups(wgs, nc, sp.ktiles, cu_slots(c, ps.twin ? 256 : 512, 9), ps.twin ? 40.0 : 25.0, ps.twin ? 0.45 : 0.40);
        }
        sp.bound = bound ? 1 : 0;
        if (bound) sp.dbg = tune(TUNE_B1_PATH) >= 16 ? (tune(TUNE_B1_PATH) >> 4) : 0;   // (tuning 12 = 16 / 32: k_bound timing ablations)
        if (bound) { sp.A = rowbuf; sp.B = colbuf; sp.a_cs = sp.b_cs = 0; }   // (the fragment-order image holds the ONE candidate in range)
        if (fast && !bound && !ps.store_out && (ps.j_mode == 0 || ps.j_mode == 2)) {
            const int h9 = sweep9_halves(sp, ps.twin, ps.epi);
            if (h9 > 0 && (long)h9 * SW9_NW <= p_zs) {       // the table allocated for the 128-tile layout holds this one
                sp.halves = h9;
                sp.rows_p_stream = sp.a_cs == 0 ? Np : Mp;
                sp.p_zs = (long)h9 * SW9_NW; sp.p_cs = sp.p_zs * ps.Z;
                nine_halves = h9;
                cgroups = choose_cgroups((long)h9 * ps.Z, nc, 1, cu_slots(c, SW9_NW == 4 ? 512 : 256, 6), 12.0, 0.9);
            }
        }
        if (fast && !ps.store_out) {
            if (const int t_ = tune(sweep2g_ok(sp) ? TUNE_CG2G : TUNE_CG2); t_ > 0) cgroups = std::max(1, std::min(nc, t_));
            if (tune(TUNE_PRINT) > 0) fprintf(stderr, "[p4v] sweep2%s tiles %d x %d z %d ktiles %d cand %d twin %d -> cgroups %d\n", sweep2g_ok(sp) ? "g" : "", sp.mtiles, sp.ntiles, ps.Z, sp.ktiles, nc, (int)ps.twin, cgroups);
        }
        CHK(launch_sweep(c, sp, ps.i8, ps.twin, ps.epi, fast || fast_cos, cgroups));
    }
    g_exec_frac = 1.0;
    if (ps.store_out) { c.ws.off = mark; return 0; }
    auto with_marks = [&](FinishParams& fp) {     // a pruned pass flags the candidates it packed into the module's plane
        if (ps.crange && pc && pc->done) { fp.mark_done = pc->done; fp.mark_n = ps.eq_n; }
        fp.crange_blk = rblk ? rblk : rblk_z;
    };
    if (nine_halves > 0) {      // k_sweep9 wrote [C][Z][halves * 8]
        const int slots = nine_halves * SW9_NW;
        FinishParams fp{part, (long)slots * ps.Z, (long)slots, slots, 1, ps.Z, slots, ps.eq_n, ps.j_mode, std::max(1, ps.j_div), ps.nj, ps.norm, scores, ps.crange};
        with_marks(fp);
        CHK(launch_finish(c, fp));
    } else if (!cosm) {
        const int gdiv = stat_ok ? s3_gw : 32;
        // k_sweep4 activation search (j_mode 0) sums the whole table; its columns are sample groups
        // (k_sweep6 skips streaming tiles that are pure padding: their table entries are never written)
        const int fin_cols = stat_ok ? (a_search ? (regs6 ? 2 * cdiv(ps.Mrows, 64) : s3_groups) : cdiv(ps.Ncols, s3_gw))
                                     : fast ? cdiv(ps.Ncols, 32) : ps.Ncols;
        FinishParams fp{part, p_cs, p_zs, NpP, stat_ok ? s3_slabs : big7 ? MT7 : MT, ps.Z, fin_cols, ps.eq_n, ps.j_mode,
                        std::max(1, (fast || stat_ok) && ps.j_mode == 1 ? cdiv(ps.j_div, gdiv) : ps.j_div), ps.nj, ps.norm, scores, ps.crange};
        with_marks(fp);
        CHK(launch_finish(c, fp));
    } else {
        // part layout [C][ZB][ZV][FS][Sp][3] with z = zb*ZV + zv
        // (k_sweep6: Z = 1, the V blocks are consecutive runs of 64-feature slabs of the one table; samples = the rows)
        // (a V block = sb_div features = sb_div / 64 whole slabs, whatever the padding behind the last block)
        const bool cosp = ps.cos6 || ps.cos7;
        const int cos6_FS = !cosp ? 0 : ps.cos_ZV == 1 ? cos6_slabs : ps.sb_div / cos_slab;
        FinishCosParams fp{part, p_cs, cosp ? (long)cos6_FS * cos6_Sp * 3 : p_zs, cosp ? cos6_Sp : Np, cosp ? cos6_FS : MT,
                           ps.cos_ZB, ps.cos_ZV, cosp ? ps.Mrows : ps.Ncols, ps.eq_n,
                           ps.cos_j_mode, std::max(1, ps.cos_j_div), ps.nj, ps.norm, scores};
        CHK(launch_finish_cos(c, fp));
    }
    if (!ps.no_select) {
        SelectParams sl{scores, ps.eq_n, ps.nj, ps.cands, ps.cand_cs, ps.cand_js, ps.cand_off, ps.interval, ps.out_js,
                        ps.out_off, ps.aux_out, ps.aux_div, ps.scores_out, ps.scores_out_ld, ps.best_out};
        CHK(launch_select(c, sl));
    }
    c.ws.off = mark;   // scratch of this pass is reusable by the next one (same stream => ordered)
    return 0;
}

// ---- exact candidate pruning ------------------------------------------------------------------------------------------------
// (the argument is with the prune kernels, p4v_kernels.h)  A pass becomes: stage A = all candidates on the HEAVIEST ~1/16 of
// the samples (by their share of the metric weight; its own small problem, scores only); B1 = the stage-A winners on all
// samples (the bound); B2 = the surviving range on all samples with the unpruned kernels, tiles, finish and selection --
// bit-identical totals for the survivors, hence the same selection (tests: every parity case runs with and without it,
// desc.reserved bit 3 / variant 4194304 switch it off).  Everything between the stages stays on the device.  Not used when the
// caller wants the full score tables, for the cosine metric (its terms are not one-signed), for fp32 operands other than the
// patch embedding's weight search, or where the sample slice would not be small against the whole.
// The sample slice of a module (see SliceCache): geometry, allocation (top of the workspace when the module keeps it, the bump
// region otherwise) and contents (ranking, share of the metric weight, gathered rows; only what changed is redone).
struct SliceGeo {
    bool lin; int segs, seg_rows, k /* rows per segment */, Ncols, K;
    const float* O; const float* G; int wt_mode; long o_ms;
    const float* row_src; long s_r, s_k; int zdiv; long s_z2, s_z;
    bool conv; PackParams conv_pk;     // the row operand is the im2col view of a conv input (flat: rows = (image, pixel))
    int k_cap;                         // rows per segment the buffers hold (>= k)
};
int slice_alloc(Ctx& c, SliceCache* sc, const SliceGeo& g, bool keep, bool bump) {
    if (sc->assigned) {
        if (sc->k != g.k_cap) return fail(P4V_ERR_INVALID, "slice cache reused with another geometry");
        return 0;
    }
    if (!keep && !bump) return 0;                 // a pass-local slice is allocated later, in the bump region
    const long zrows = (long)g.segs * g.seg_rows, out_elems = (long)g.segs * g.k_cap * g.Ncols, row_elems = (long)g.segs * g.k_cap * g.K;
    if (keep) {
        sc->idx = reinterpret_cast<int*>(c.ws.get_top((size_t)g.segs * g.k_cap * sizeof(int)));
        sc->mass = reinterpret_cast<float*>(c.ws.get_top((size_t)zrows * sizeof(float)));
        sc->Os = reinterpret_cast<float*>(c.ws.get_top((size_t)out_elems * sizeof(float)));
        sc->Gs = reinterpret_cast<float*>(c.ws.get_top((size_t)out_elems * sizeof(float)));
        sc->Rs = reinterpret_cast<float*>(c.ws.get_top((size_t)row_elems * sizeof(float)));
        sc->frac = reinterpret_cast<float*>(c.ws.get_top(256));
        sc->assigned = true;
    } else {
        sc->idx = c.ws.get<int>((size_t)g.segs * g.k_cap);
        sc->mass = c.ws.get<float>((size_t)zrows);
        sc->Os = c.ws.get<float>((size_t)out_elems);
        sc->Gs = c.ws.get<float>((size_t)out_elems);
        sc->Rs = c.ws.get<float>((size_t)row_elems);
    }
    sc->k = g.k_cap;
    return 0;
}
int slice_fill(Ctx& c, SliceCache* sc, const SliceGeo& g, bool host_sync_ok) {
    const long zrows = (long)g.segs * g.seg_rows;
    const void* wsrc = g.G ? (const void*)g.G : (const void*)g.O;
    const bool new_idx = sc->idx_src != wsrc || sc->idx_wt != g.wt_mode || (g.wt_mode != 1 && sc->o_src != g.O);
    if (new_idx) {
        // the heaviest rows of every segment by their share of the metric weight
        CHK(enqueue(c, KERN(RowMassParams, k_row_mass), dim3((unsigned)cdiv(zrows, 4)), dim3(256), 0, RowMassParams{g.G ? g.G : g.O, g.O, zrows, (long)g.Ncols, g.wt_mode, sc->mass}));
        CHK(enqueue(c, KERN(TopkParams, k_topk_rows), dim3(g.segs), dim3(1024), 0, TopkParams{sc->mass, g.seg_rows, g.k, sc->idx}));
        sc->idx_src = wsrc; sc->idx_wt = g.wt_mode;
        sc->o_src = sc->g_src = sc->r_src = nullptr;
        if (sc->frac && host_sync_ok && !(g_variant & 8388608)) {
            // once per module: is the slice worth it?  The bounds are as tight as the share of the metric weight the slice
            // holds (ViT class-token rows: > 0.99; the qkv layers, whose keys and values of every token feed the class token:
            // 0.72 -- still worth it, measured: 187 -> 168 ms per ViT-B calibration); below 0.5 most candidates survive and
            // the three stages cost more than the full sweep they replace -- such a module keeps the full sweep (variant
            // 8388608: always prune).  Round 6: that was measured on 6 304-row layers.  On the 128-image configurations a pass
            // is 10-200 x larger while the stages' fixed costs are the same, and Swin (no class token: the slice holds 0.1-0.25
            // of the weight) still loses a fifth of its search time to candidates the slice already rules out: from 65 536
            // sample rows on a module prunes when its slice holds 5 % (Swin-B/384 x 128: search 4.75 -> 3.83 s; 25 %: no change,
            // 1 %: 3.81 s).  What survives is re-evaluated exactly as before: the selection does not depend on the threshold.
            float f = 1.0f;
            int* hm = tune(TUNE_B1_PATH) == 8 ? nullptr : host_mirror(c);
            CHK(enqueue(c, KERN(MassFracParams, k_mass_fraction), dim3(1), dim3(1024), 0,
                        MassFracParams{sc->mass, zrows, sc->idx, g.segs, g.seg_rows, g.k, sc->frac, hm ? reinterpret_cast<float*>(hm + 4) : nullptr}));
            if (!hm) CHK(q_d2h(c, &f, sc->frac, sizeof f));
            CHK(q_sync(c));
            if (hm) f = *reinterpret_cast<volatile float*>(hm + 4);
            if (tune(TUNE_PRINT) > 0) fprintf(stderr, "[p4v] slice holds %.4f of the metric weight (%d x %d of %d rows)\n", f, g.segs, g.k, g.seg_rows);
            sc->frac_host = f;
            if (g.k < g.k_cap && !(f >= 0.97f)) {      // the small slice does not hold the weight: the full one (ranked again)
                SliceGeo g2 = g;
                g2.k = g.k_cap;
                sc->idx_src = nullptr;
                return slice_fill(c, sc, g2, host_sync_ok);
            }
            const bool big_pass = zrows >= (tune(TUNE_LOOSE_ROWS) > 0 ? (long)tune(TUNE_LOOSE_ROWS) : 65536L);
            if (!(f >= (tune(TUNE_LOOSE_PCT) > 0 ? 0.01f * tune(TUNE_LOOSE_PCT) : big_pass ? 0.05f : 0.5f))) { sc->loose = true; return 0; }
        }
        sc->k_eff = g.k;
    }
    const int rows = g.segs * g.k;
    int gather_rc = 0;
    auto gather = [&](const float* src, long s0, long s3, int d3, float* dst, int seg, int zdiv, long sz2, long sz) {
        GatherParams gp{src, s0, 0, 0, s3, 1, 1, d3, sc->idx, rows, dst, seg, zdiv, sz2, sz};
        const long total = (long)rows * d3;
        if (const int r_ = enqueue(c, KERN(GatherParams, k_gather), dim3((unsigned)std::min<long>(cdiv(total, 256), 256L * 16)), dim3(256), 0, gp)) gather_rc = r_;
    };
    const int seg = g.lin ? 0 : g.k;
    const long o_seg = (long)g.seg_rows * g.Ncols;                // raw_out / raw_grad: dense [Z][M][N]
    if (sc->o_src != g.O) { gather(g.O, g.o_ms, 1, g.Ncols, sc->Os, seg, 1, o_seg, 0); sc->o_src = g.O; }
    if (g.G && sc->g_src != g.G) { gather(g.G, g.o_ms, 1, g.Ncols, sc->Gs, seg, 1, o_seg, 0); sc->g_src = g.G; }
    if (sc->r_src != g.row_src) {
        if (g.conv) {
            const long total = (long)rows * g.K;
            CHK(enqueue(c, KERN(GatherIm2colParams, k_gather_im2col), dim3((unsigned)std::min<long>(cdiv(total, 256), 256L * 16)), dim3(256), 0, GatherIm2colParams{g.conv_pk, sc->idx, rows, sc->Rs}));
        } else if (g.lin) gather(g.row_src, g.s_r, 1, g.K, sc->Rs, 0, 1, 0, 0);
        else gather(g.row_src, g.s_r, g.s_k, g.K, sc->Rs, g.k, g.zdiv, g.s_z2, g.s_z);
        sc->r_src = g.row_src;
        sc->aplane.valid = false;
    }
    return gather_rc;
}

bool prune_ok(const Pass& ps) {
    if (!ps.prunable || !(ps.i8 || ps.prunable_f32) || ps.epi == EPI_COS || ps.store_out || ps.scores_out || ps.best_out || ps.crange || ps.no_select) return false;
    if ((g_variant & 4194304) || ps.eq_n < 32 || ps.nj < 1 || ps.nj > 4096) return false;
    return true;
}
int launch_pass_select(Ctx& c, const Pass& ps, const float* scores) {
    SelectParams sl{scores, ps.eq_n, ps.nj, ps.cands, ps.cand_cs, ps.cand_js, ps.cand_off, ps.interval, ps.out_js,
                    ps.out_off, ps.aux_out, ps.aux_div, ps.scores_out, ps.scores_out_ld, ps.best_out};
    return launch_select(c, sl);
}

#define PRUNE_COUNT(i) do { if (!c.dry) g_prune_cnt[i].fetch_add(1, std::memory_order_relaxed); } while (0)
// The margin by which a stage-A (slice) score must lie below the bound L* before the candidate is dropped.  Both numbers are
// sums of same-signed terms whose per-element values are computed by the same arithmetic in every stage; what differs is the
// order of summation.  Every sweep sums in fp32 only INSIDE a wave -- at most PRUNE_FP32_CHAIN additions in a row: the 128
// outputs a lane of k_sweep7 owns per candidate (the longest chain of any sweep; k_sweep6: 32, k_sweep2/8/9: 32-48, the generic
// and k_sos_split sweeps: 64) + the 6 levels of the cross-lane reduction -- and everything across waves / tiles / slabs is
// k_finish's fp64.  A chain of n additions of same-signed fp32 terms is off by at most (n - 1) u / (1 - (n - 1) u) relative,
// u = 2^-24; the slice sum and the bound each carry one such error, so 2 n u = 1.6e-5 bounds their disagreement and the margin
// is 4 x that, never below the 1e-4 of round 3 (which is therefore what is used; a kernel with a longer fp32 chain must raise
// PRUNE_FP32_CHAIN).  Variant 134217728 cross-checks every pruned pass against the full sweep (tests).
constexpr int PRUNE_FP32_CHAIN = 128 + 6;
inline float prune_margin() { return std::max(1e-4f, 4.0f * 2.0f * (float)PRUNE_FP32_CHAIN * 5.9604645e-8f); }
int prune_crosscheck_begin(Ctx& c, const float* interval, int n, std::vector<float>& keep) {
    keep.resize(n);
    CHK(q_d2h(c, keep.data(), interval, sizeof(float) * n));
    CHK(q_sync(c));
    return 0;
}
int prune_crosscheck_end(Ctx& c, const float* interval, int n, const std::vector<float>& pruned, const char* what) {
    std::vector<float> full(n);
    CHK(q_d2h(c, full.data(), interval, sizeof(float) * n));
    CHK(q_sync(c));
    for (int i = 0; i < n; ++i)
        if (std::memcmp(&full[i], &pruned[i], sizeof(float)) != 0)
            return fail(P4V_ERR_INVALID, "exact candidate pruning selected another candidate than the full sweep (%s, output %d: %.9g vs %.9g)", what, i, (double)pruned[i], (double)full[i]);
    return 0;
}
// ---- stage A of a pruned MatMul B search in one kernel (k_slice_b: B quantised in the kernel, no candidate planes) ---------------
// `a` is the stage-A pass run_pass_pruned built (row operand = the 16-row slices, dense [Z][16][K]; column operand = the whole B,
// candidate-expanded); `SA` receives the scores [eq_n][nj].  Conditions: int8, head-wise scales (block = z % H on both the
// quantiser and the output scales), K <= 256, N <= 208, no bias, a difference metric.
bool slice_b_ok(const Pass& a) {
    const PackParams& b = a.col.pk;
    return a.i8 && a.col.expanded && !a.row.expanded && !(a.twin && a.row2.expanded) && a.Z > 1 && a.Mrows <= 16 && !a.store_out &&
           a.epi != EPI_COS && a.epi != EPI_STORE && a.epi != EPI_FWD && !a.bias && a.use_s1 && a.sb_mode == 2 && a.j_mode == 2 &&
           a.sb_div == a.j_div && a.s_cs == a.sb_div && !a.crange && a.scores_keep && a.no_select &&
           !b.conv && b.mode == PACK_SYM && b.blk_mode == 2 && b.blk_div == a.sb_div && b.scales && !a.col_zs_shared && !a.row_zs_shared &&
           rup(a.K, 64) <= 256 && a.Ncols <= 208 && (rup(a.K, 64) == 64 || a.Ncols <= 64) && a.o_ms == a.Ncols && a.o_ns == 1 &&
           a.o_zs == (long)a.Mrows * a.Ncols && !a.o_bs && !a.o_nbs && !g_force_v1 && tune(TUNE_B1_PATH) != 6;
}
int run_slice_b(Ctx& c, Pass& a, float* SA) {
    const int Kp = (int)rup(a.K, 64), Z = a.Z;
    const size_t mark = c.ws.off;
    int8_t* A1 = c.ws.get<int8_t>((size_t)Z * 16 * Kp);
    int8_t* A2 = a.twin ? c.ws.get<int8_t>((size_t)Z * 16 * Kp) : nullptr;
    const bool v2 = tune(TUNE_B1_PATH) != 10;           // k_slice_b2 (B in registers, waves deal the column blocks); 12 = 10: k_slice_b (A/B)
    float* part = c.ws.get<float>((size_t)a.eq_n * Z * (v2 ? 4 : 1));
    float* S1 = a.S1_pre ? a.S1_pre : c.ws.get<float>((size_t)a.eq_n * a.s_cs);
    float* S2 = !a.twin ? nullptr : a.S2_pre ? a.S2_pre : c.ws.get<float>((size_t)a.eq_n * a.s_cs);
    if (!c.ws.ok()) return fail(P4V_ERR_WORKSPACE, "workspace too small: need >= %zu bytes", c.ws.off);
    if (!a.s_ready) {
        a.s1.S = S1; a.s1.C = a.eq_n; a.s1.nblk = a.s_cs;
        CHK(launch_scale(c, a.s1));
        if (a.twin) { a.s2.S = S2; a.s2.C = a.eq_n; a.s2.nblk = a.s_cs; CHK(launch_scale(c, a.s2)); }
    }
    auto pack16 = [&](const Operand& op, int8_t* dst) -> int {     // the slice's fixed plane(s): [Z][16][Kp], row-major
        PackParams pk = op.pk;
        pk.Rp = 16; pk.Kp = Kp; pk.dst = dst; pk.Z = Z; pk.C = 1; pk.c_inner = 0;
        return launch_pack<int8_t>(c, pk);
    };
    CHK(pack16(a.row, A1));
    if (a.twin) CHK(pack16(a.row2, A2));
    if (!c.dry) {
        const PackParams& b = a.col.pk;
        SliceBParams kp{};
        kp.A = A1; kp.A2 = A2;
        kp.B = b.src; kp.b_z2 = b.s_z2; kp.b_z = b.s_z; kp.b_n = b.s_r; kp.b_k = b.s_k; kp.zdiv = b.zdiv;
        kp.bscale = b.scales; kp.bs_cs = b.sc_cs; kp.bs_div = b.blk_div; kp.lo = b.lo; kp.hi = b.hi;
        kp.S1 = S1; kp.S2 = S2; kp.s_cs = a.s_cs; kp.s_div = a.sb_div;
        kp.O = a.O; kp.Wt = a.G ? a.G : a.O; kp.wt_mode = a.wt_mode;
        kp.Z = Z; kp.M = a.Mrows; kp.K = a.K; kp.Kp = Kp; kp.N = a.Ncols; kp.C = a.eq_n; kp.part = part;
        const int nb = cdiv(a.Ncols, 16);
        const size_t lds = (size_t)nb * 16 * (Kp + 4) * sizeof(float);
        // k_slice_b: >= 1024 workgroups, >= 4 candidates each (one per wave); k_slice_b2: every wave runs every candidate of its
        // workgroup, the prologue (B -> registers) is paid per workgroup: >= 512 workgroups of >= 10 candidates
        // (k_slice_b2 keeps 3 workgroups per CU -- 2 with the twin's second accumulator set, 250 registers: whole rounds of 256 x that)
        const int slots = std::max(8, 256 * (a.twin ? 2 : 3) / lockstep(c, 13));
        int g2 = std::max(1, std::min(a.eq_n / 10, cdiv(std::max(1, 512 / lockstep(c, 13)), Z)));
        // no mostly-empty last round -- where the rounds are few (ViT: 384 batch entries; with tens of thousands of them, Swin's
        // windows, the tail does not matter and more groups only repeat the prologue)
        while ((long)Z * g2 < 4L * slots && g2 < a.eq_n / 10 && ((long)Z * g2) % slots != 0 && ((long)Z * g2) % slots < slots * 3 / 4) ++g2;
        if (tune(TUNE_CG2) > 0) g2 = std::max(1, std::min(a.eq_n, tune(TUNE_CG2)));
        const int groups = v2 ? g2 : std::max(1, std::min(a.eq_n / 4, cdiv(std::max(1, 1024 / lockstep(c, 13)), Z)));
        const dim3 grid(Z, groups), block(256);
        const float qbias = (v2 && b.lo == -128 && b.hi == 127 && tune(TUNE_B1_PATH) != 11) ? cvt_bias(c) : 0.0f;   // 12 = 11: quant_fast1 in k_slice_b2 (A/B)
        const StatInfo si{13, 16.0 * nb * 16 * Kp * Z * a.eq_n * (a.twin ? 2 : 1), (double)a.Mrows * a.Ncols * a.K * Z * a.eq_n, g_stage, Z, groups,
                          4.0 * ((double)a.Mrows * a.K * Z + (double)a.Ncols * a.K * Z) + (a.G ? 8.0 : 4.0) * (double)a.Mrows * a.Ncols * Z};
        const StatInfo* sp = &si;
        int r_ = 0;
        if (v2) {
            SliceB2Params kp2{kp, qbias};
#define P4V_LAUNCH_SB2(TW, KTM, NBW)                                                                                                  \
            P4V_EPI4(a.epi, r_ = (qbias != 0.0f) ? enqueue(c, KERN_T(SliceB2Params, k_slice_b2, TW, KTM, NBW, E, true), grid, block, 0, kp2, sp)   \
                                                 : enqueue(c, KERN_T(SliceB2Params, k_slice_b2, TW, KTM, NBW, E, false), grid, block, 0, kp2, sp); break)
            if (Kp == 64) { if (a.twin) { P4V_LAUNCH_SB2(true, 1, 4) } else { P4V_LAUNCH_SB2(false, 1, 4) } }
            else { if (a.twin) { P4V_LAUNCH_SB2(true, 4, 1) } else { P4V_LAUNCH_SB2(false, 4, 1) } }
#undef P4V_LAUNCH_SB2
        } else {
#define P4V_LAUNCH_SB(TW, KTM, NBM) P4V_EPI4(a.epi, r_ = enqueue(c, KERN_T(SliceBParams, k_slice_b, TW, KTM, NBM, E), grid, block, lds, kp, sp); break)
            if (Kp == 64) { if (a.twin) { P4V_LAUNCH_SB(true, 1, 13) } else { P4V_LAUNCH_SB(false, 1, 13) } }
            else { if (a.twin) { P4V_LAUNCH_SB(true, 4, 4) } else { P4V_LAUNCH_SB(false, 4, 4) } }
#undef P4V_LAUNCH_SB
        }
        if (r_) return r_;
    }
    const int pw = v2 ? 4 : 1;                          // floats per (candidate, batch entry): one per wave of k_slice_b2
    FinishParams fp{part, (long)Z * pw, (long)pw, pw, 1, Z, pw, a.eq_n, a.j_mode, std::max(1, a.j_div), a.nj, a.norm, SA, nullptr};
    CHK(launch_finish(c, fp));
    c.ws.off = mark;
    return 0;
}

// ... and of a pruned MatMul A search with K <= 64 (k_slice_a: the slice is quantised per candidate in registers, B fixed)
bool slice_a_ok(const Pass& a) {
    const PackParams& r = a.row.pk;
    return a.i8 && a.row.expanded && !a.col.expanded && !a.twin && a.Z > 1 && a.Mrows <= 16 && !a.store_out &&
           a.epi != EPI_COS && a.epi != EPI_STORE && a.epi != EPI_FWD && !a.bias && a.use_s1 && a.sb_mode == 2 && a.j_mode == 2 &&
           a.sb_div == a.j_div && a.s_cs == a.sb_div && !a.crange && a.scores_keep && a.no_select &&
           !r.conv && r.mode == PACK_SYM && r.blk_mode == 2 && r.blk_div == a.sb_div && r.scales && r.s_k == 1 && r.s_r == a.K &&
           r.zdiv > 0 && r.s_z == (long)a.Mrows * a.K && r.s_z2 == (long)r.zdiv * a.Mrows * a.K && a.Mrows == 16 &&
           !a.col_zs_shared && !a.row_zs_shared && a.K <= 64 && a.Ncols <= 208 && a.o_ms == a.Ncols && a.o_ns == 1 &&
           a.o_zs == (long)a.Mrows * a.Ncols && !a.o_bs && !a.o_nbs && !g_force_v1 && tune(TUNE_B1_PATH) != 6;
}
int run_slice_a(Ctx& c, Pass& a, float* SA) {
    const int Z = a.Z, nb = cdiv(a.Ncols, 16);
    const size_t mark = c.ws.off;
    int8_t* Bp = c.ws.get<int8_t>((size_t)Z * nb * 16 * 64);
    float* part = c.ws.get<float>((size_t)a.eq_n * Z);
    float* S1 = a.S1_pre ? a.S1_pre : c.ws.get<float>((size_t)a.eq_n * a.s_cs);
    if (!c.ws.ok()) return fail(P4V_ERR_WORKSPACE, "workspace too small: need >= %zu bytes", c.ws.off);
    if (!a.s_ready) {
        a.s1.S = S1; a.s1.C = a.eq_n; a.s1.nblk = a.s_cs;
        CHK(launch_scale(c, a.s1));
    }
    {
        PackParams pk = a.col.pk;                          // the fixed column operand: [Z][nb * 16][64], row-major
        pk.Rp = nb * 16; pk.Kp = 64; pk.dst = Bp; pk.Z = Z; pk.C = 1; pk.c_inner = 0;
        CHK(launch_pack<int8_t>(c, pk));
    }
    if (!c.dry) {
        const PackParams& r = a.row.pk;
        SliceAParams kp{};
        kp.A = r.src; kp.B = Bp;
        kp.ascale = r.scales; kp.as_cs = r.sc_cs; kp.as_div = r.blk_div; kp.lo = r.lo; kp.hi = r.hi;
        kp.S1 = S1; kp.s_cs = a.s_cs; kp.s_div = a.sb_div;
        kp.O = a.O; kp.Wt = a.G ? a.G : a.O; kp.wt_mode = a.wt_mode;
        kp.Z = Z; kp.M = a.Mrows; kp.K = a.K; kp.N = a.Ncols; kp.C = a.eq_n; kp.part = part;
        kp.qbias = (r.lo == -128 && r.hi == 127 && tune(TUNE_B1_PATH) != 11) ? cvt_bias(c) : 0.0f;
        const int groups = std::max(1, std::min(a.eq_n / 4, cdiv(std::max(1, 1024 / lockstep(c, 14)), Z)));
        const dim3 grid(Z, groups), block(256);
        const StatInfo si{14, 16.0 * nb * 16 * 64 * Z * a.eq_n, (double)a.Mrows * a.Ncols * a.K * Z * a.eq_n, g_stage, Z, groups,
                          4.0 * ((double)a.Mrows * a.K * Z + (double)a.Ncols * a.K * Z) + (a.G ? 8.0 : 4.0) * (double)a.Mrows * a.Ncols * Z};
        const StatInfo* sp = &si;
        P4V_EPI4(a.epi, CHK(enqueue(c, KERN_T(SliceAParams, k_slice_a, 13, E), grid, block, 0, kp, sp)); break)
    }
    FinishParams fp{part, (long)Z, 1L, 1, 1, Z, 1, a.eq_n, a.j_mode, std::max(1, a.j_div), a.nj, a.norm, SA, nullptr};
    CHK(launch_finish(c, fp));
    c.ws.off = mark;
    return 0;
}

int run_pass_pruned_impl(Ctx& c, Pass& ps);
int run_pass_pruned(Ctx& c, Pass& ps) {
    if (!(g_variant & 134217728) || !prune_ok(ps) || (!c.dry && !ps.interval)) return run_pass_pruned_impl(c, ps);
    // debug cross-check: the pruned pass, then the full sweep of the same pass; the selections must be bit-identical
    // (the dry run plans both: the full sweep's planes live in the bump region)
    const int n = ps.out_off + (std::max(1, ps.nj) - 1) * ps.out_js + 1;
    std::vector<float> keep;
    CHK(run_pass_pruned_impl(c, ps));
    if (!c.dry) CHK(prune_crosscheck_begin(c, ps.interval, n, keep));
    Pass full = ps;
    full.prunable = false; full.scache = nullptr; full.scache2 = nullptr; full.cache = nullptr; full.ecache = nullptr;
    CHK(run_pass(c, full));
    return c.dry ? 0 : prune_crosscheck_end(c, ps.interval, n, keep, "search pass");
}
int run_pass_pruned_impl(Ctx& c, Pass& ps) {
    if (!prune_ok(ps)) { PRUNE_COUNT(3); return run_pass(c, ps); }
    const bool lin = ps.Z == 1;
    // geometry of the slice.  Linear: the k heaviest of its M samples (rows of x / raw_out / raw_grad).  MatMul: the 16
    // heaviest rows (queries) of EVERY batch entry (image, head) -- the column operand stays whole.  Measured on ViT-B/224 x 32
    // (tools/row_mass.py, tools/row_mass_mm.py): 99.9 % of raw_grad^2 sits in 1/8 of a Linear's samples and 99.7-100 % in ONE
    // row of each (image, head) of the attention matmuls -- the class-token rows, the only ones the classifier reads.
    const int segs = lin ? 1 : ps.Z;                                 // ranking segments
    const int seg_rows = ps.Mrows;                                   // rows ranked per segment
    int k;                                                           // rows taken per segment
    if (lin) {
        k = (int)std::min<long>(rup(std::max(1, ps.Mrows / (tune(TUNE_SLICE_DIV) > 0 ? tune(TUNE_SLICE_DIV) : 16)), 256), rup(ps.Mrows, 256));
        if ((long)k * 5 > (long)ps.Mrows * 2) { PRUNE_COUNT(2); return run_pass(c, ps); }   // slice > 40 % of the samples: not worth the stages
        // dense row-major operands only (or the im2col rows of a conv input, gathered element by element)
        const bool conv_rows = ps.row.pk.conv && ps.row.pk.Z == 1 && !ps.twin;
        if (ps.o_ms != ps.Ncols || ps.o_ns != 1 || ps.o_bs || ps.o_nbs || ps.row.pk.zdiv > 0 ||
            (!conv_rows && (ps.row.pk.conv || ps.row.pk.s_k != 1 || ps.row.pk.s_r != ps.K))) { PRUNE_COUNT(2); return run_pass(c, ps); }
    } else {
        k = 16;
        if (ps.Mrows < 64 || ps.row_zs_shared || ps.o_zs != (long)ps.Mrows * ps.Ncols || ps.o_ms != ps.Ncols || ps.o_ns != 1 ||
            ps.o_bs || ps.o_nbs || ps.row.pk.zdiv <= 0 || ps.row.pk.conv) { PRUNE_COUNT(2); return run_pass(c, ps); }
    }
    SliceCache local;
    SliceCache* sc = ps.scache ? ps.scache : &local;
    if (sc->loose) { PRUNE_COUNT(2); return run_pass(c, ps); }
    // A Linear first tries a QUARTER of the slice: the M/64 (at least 128) heaviest samples -- in a ViT the class-token rows, one
    // per image of 197+ tokens, are among them -- hold > 97 % of the weight in every layer but qkv, and stage A costs in proportion
    // to the slice.  Measured (ViT-B/224 x 32, one box): full slice 155.4 ms per calibration, 256 rows 151.2, 128 rows 148.8,
    // 64 rows 148.6 (variant 67108864: always the full slice; p4v_debug_set_tuning(11, rows) overrides the size)
    const int k_cap = k;
    if (sc->k_eff > 0) k = sc->k_eff;
    else if (lin && ps.scache && ps.host_sync_ok && !c.dry && k_cap >= 512 && !(g_variant & (8388608 | 67108864)))
        k = tune(TUNE_SLICE_SMALL) > 0 ? std::min(tune(TUNE_SLICE_SMALL), k_cap)
                                       : (int)std::min<long>(std::max<long>(128, rup(ps.Mrows / 64, 64)), k_cap / 2);
    SliceGeo geo{lin, segs, seg_rows, k, ps.Ncols, ps.K, ps.O, ps.G, ps.wt_mode, ps.o_ms, ps.row.pk.src, ps.row.pk.s_r, ps.row.pk.s_k,
                 ps.row.pk.zdiv, ps.row.pk.s_z2, ps.row.pk.s_z, lin && ps.row.pk.conv != 0, ps.row.pk, k_cap};
    CHK(slice_alloc(c, sc, geo, ps.scache != nullptr, /*bump=*/false));
    // Second tier (Linear layers whose weight is spread over more samples than the first slice holds -- the qkv layers: the keys
    // and values of EVERY token feed the class token, 0.72 of the weight in their 512 heaviest samples, ~32 of 100 candidates
    // survive stage B1's bound and stage B2 swept them over all 6304 samples: 37 % of the sweep time of a ViT-B calibration):
    // the survivors are first swept over the M/4 heaviest samples (a partial sum over ANY subset of the samples is an upper
    // bound; this one is ~3 x tighter) and only what survives THAT goes to stage B2.  The buffers are planned unconditionally
    // (the dry run cannot know whether a module will need them).
    SliceCache* sc2 = (lin && ps.scache && ps.scache2 && !ps.row.pk.conv && tune(TUNE_TIER2) != 1) ? ps.scache2 : nullptr;
    // (M / 4: measured on ViT-B/224 x 32, k_sweep6 time per calibration 30.6 ms without the tier, 29.1 / 28.1 / 28.7 / 29.4 ms with M / 3, 4, 6, 8)
    const int k2 = (int)rup(std::max(1, ps.Mrows / (tune(TUNE_TIER2_DIV) > 0 ? tune(TUNE_TIER2_DIV) : 4)), 256);
    if (sc2 && (k2 < 2 * k_cap || (long)k2 * 5 > (long)ps.Mrows * 2)) sc2 = nullptr;
    SliceGeo geo2 = geo;
    geo2.k = geo2.k_cap = k2;
    if (sc2) CHK(slice_alloc(c, sc2, geo2, true, /*bump=*/false));
    const size_t mark = c.ws.off;
    const size_t tab = (size_t)ps.eq_n * std::max(1, ps.nj);
    float* SA = c.ws.get<float>(tab);
    float* SB = c.ws.get<float>(tab);
    float* S2 = c.ws.get<float>(tab);
    float* SA2 = sc2 ? c.ws.get<float>(tab) : nullptr;
    int* r1 = c.ws.get<int>(6);
    int* r2 = r1 + 2;
    int* r3 = r1 + 4;
    int* best_idx = c.ws.get<int>((size_t)std::max(1, ps.nj));
    int* rblk2 = c.ws.get<int>((size_t)4 * std::max(1, ps.nj));      // per-score-block survivor ranges of the hull (r2) ...
    int* rblk3 = rblk2 + 2 * std::max(1, ps.nj);                       // ... and of the second tier's (r3)
    float* vrow = c.ws.get<float>((size_t)std::max(1, ps.cand_cs));
    float* S1s = ps.use_s1 ? c.ws.get<float>((size_t)ps.eq_n * ps.s_cs) : nullptr;             // one scale table for all stages
    float* S2s = (ps.use_s1 && ps.twin) ? c.ws.get<float>((size_t)ps.eq_n * ps.s_cs) : nullptr;
    if (!ps.scache) CHK(slice_alloc(c, sc, geo, false, /*bump=*/true));
    if (!c.ws.ok()) return fail(P4V_ERR_WORKSPACE, "workspace too small: need >= %zu bytes", c.ws.off);
    float *Os = sc->Os, *Gs = ps.G ? sc->Gs : nullptr, *Rs = sc->Rs;
    Pass a = ps;
    if (!c.dry) {
        CHK(slice_fill(c, sc, geo, ps.host_sync_ok));
        if (sc->loose) { c.ws.off = mark; PRUNE_COUNT(2); return run_pass(c, ps); }
        if (sc->k_eff > 0) k = sc->k_eff;
    }
    PRUNE_COUNT(0);
    // stage A: all candidates on the slice
    a.O = Os; a.G = ps.G ? Gs : nullptr;
    a.Mrows = k;
    auto sliced = [&](PackParams& pk) {
        pk.src = Rs; pk.R = k; pk.s_k = 1; pk.s_r = ps.K; pk.conv = 0;
        if (!lin) { pk.s_z = (long)k * ps.K; pk.s_z2 = (long)pk.zdiv * k * ps.K; }
    };
    sliced(a.row.pk);
    if (ps.twin) sliced(a.row2.pk);
    if (!lin) a.o_zs = (long)k * ps.Ncols;
    // the candidate-expanded plane of the module is shared with stage A when the slice leaves that operand whole (the column
    // operand: weights of a Linear, B of a matmul): stage A packs all candidates into the module's plane cache once, the later
    // stages and rounds find them there -- as the unpruned passes do
    // ... and when it is the sliced operand that is expanded (activation search), its slice plane is kept with the slice
    a.cache = ps.col.expanded ? ps.cache : (ps.cache && ps.scache) ? &sc->aplane : nullptr;
    a.ecache = nullptr; a.scores_keep = SA; a.no_select = true;
    a.S1_pre = S1s; a.S2_pre = S2s; a.s_ready = false;
    // several score blocks whose entries of the candidate table are exactly one row: stage B1 on ONE synthetic candidate
    const bool virt = ps.nj > 1 && ps.cand_off == 0 && ps.cand_js * ps.nj == ps.cand_cs && ps.cand_cs <= 4096 && !(g_variant & 16777216);
    PruneParams pp{SA, SB, ps.eq_n, ps.nj, prune_margin(), r1, r1, virt ? 1 : 0, best_idx, ps.cands, ps.cand_cs, ps.cand_js, ps.cand_off, vrow};
    g_stage = 1;
    { const int r_ = slice_b_ok(a) ? run_slice_b(c, a, SA) : slice_a_ok(a) ? run_slice_a(c, a, SA) : run_pass(c, a); g_stage = 0; if (r_) return r_; }
    CHK(enqueue(c, KERN(PruneParams, k_prune_pick), dim3(1), dim3(256), 0, pp));
    // stage B1: the stage-A winners on all samples -> the bound
    Pass b1 = ps;
    b1.scores_keep = SB; b1.no_select = true;
    if (!virt) { b1.S1_pre = S1s; b1.S2_pre = S2s; b1.s_ready = true; }
    if (virt) {
        b1.eq_n = 1; b1.cache = nullptr;
        auto swap = [&](const float*& ptr) { if (ptr == ps.cands) ptr = vrow; };
        swap(b1.row.pk.scales); swap(b1.row2.pk.scales); swap(b1.col.pk.scales);
        swap(b1.s1.x); swap(b1.s1.y); swap(b1.s2.x); swap(b1.s2.y);
        b1.cands = vrow;
    } else b1.crange = r1;
    // ONE candidate per score block over all samples: the kernel built for that (k_bound; tuning 12=3 keeps the sweep kernels).
    // Its totals are summed in another order than the sweeps', so from here on nothing may depend on stage B1's numbers but the
    // bound: the selections below are either "the only survivor of every block" (no totals involved) or come from stage B2,
    // which always re-evaluates stage B1's candidates together with the other survivors.  The candidate's plane is packed for
    // this pass (one candidate: ~12 us) instead of being read from the module's plane, whose layout belongs to its sweep kernel.
    if (lin && ps.i8 && !ps.twin && (virt || ps.nj == 1) && tune(TUNE_B1_PATH) != 3) { b1.bound_kernel = true; b1.cache = nullptr; b1.ecache = nullptr; }
    g_stage = 2;
    { const int r_ = run_pass(c, b1); g_stage = 0; if (r_) return r_; }
    // the survivors, and -- when there are none besides stage B1's candidates -- the pass's selection from its totals
    pp.r_out = r2; pp.rblk = rblk2;
    int* hm = (ps.host_sync_ok && !c.dry && tune(TUNE_B1_PATH) != 8) ? host_mirror(c) : nullptr;
    pp.r_host = hm; pp.rblk_host = hm ? hm + MIR_RB1 : nullptr;
    int hblk[2 * MIR_BLK] = {};                             // host copy of the per-block ranges stage A2 / B2 run on (nj <= MIR_BLK)
    int hlo = 0, hhi = 0;
    bool hblk_ok = false;
    const bool hull_selects = !ps.scores_out && ps.interval && (virt || ps.nj <= 32);   // (its non-virt selection is serial over the blocks)
    SelectParams hsl{SB, ps.eq_n, ps.nj, ps.cands, ps.cand_cs, ps.cand_js, ps.cand_off, hull_selects ? ps.interval : nullptr,
                     ps.out_js, ps.out_off, ps.aux_out, ps.aux_div, nullptr, 0, ps.best_out};
    attach_mirror(hsl);
    CHK(enqueue(c, KERN(HullParams, k_prune_hull), dim3(1), dim3(256), 0, HullParams{pp, hsl}));
    // nothing survives besides stage B1's candidates: the pass's selection without another sweep
    auto select_without_b2 = [&]() -> int {
        PRUNE_COUNT(1);
        if (hull_selects) {}          // k_prune_hull made the selection
        else if (virt) {          // every block's only survivor is its stage-A winner: the table with those entries filled in
            CHK(enqueue(c, KERN(FillParams, k_fill_f32), dim3(cdiv((long)tab, 256)), dim3(256), 0, FillParams{S2, -INFINITY, (int)tab}));
            CHK(enqueue(c, KERN(MergeVirtParams, k_merge_virtual), dim3(cdiv(ps.nj, 64)), dim3(64), 0, MergeVirtParams{S2, SB, best_idx, ps.nj}));
            CHK(launch_pass_select(c, ps, S2));
        } else CHK(launch_pass_select(c, ps, SB));
        c.ws.off = mark;
        return 0;
    };
    int nsurv = 0;
    if (ps.host_sync_ok && !c.dry) {
        // the caller synchronises after this pass anyway: read the survivor range (8 bytes); in the usual case stage B1's
        // candidates are the only survivors and its totals decide -- the ~10 launches of an empty stage B2 are not made
        int h[2] = {0, 1};
        if (!hm) CHK(q_d2h(c, h, r2, sizeof h));
        CHK(q_sync(c));
        if (hm) { h[0] = reinterpret_cast<volatile int*>(hm)[0]; h[1] = reinterpret_cast<volatile int*>(hm)[1]; }
        if (h[0] >= h[1]) return select_without_b2();
        nsurv = h[1] - h[0];
        hlo = h[0]; hhi = h[1];
        if (hm && ps.nj <= MIR_BLK) { for (int i = 0; i < 2 * ps.nj; ++i) hblk[i] = reinterpret_cast<volatile int*>(hm)[MIR_RB1 + i]; hblk_ok = true; }
    }
    // second tier: many survivors of a slice that holds well under all of the weight -> sweep THEM over the larger slice first
    const int* rB = r2;
    const int* rBblk = rblk2;
    const int t2min = tune(TUNE_TIER2) >= 2 ? tune(TUNE_TIER2) : 8;
    if (sc2 && (c.dry || (ps.host_sync_ok && nsurv >= t2min && sc->frac_host < 0.9f))) {
        if (!c.dry) CHK(slice_fill(c, sc2, geo2, /*host_sync_ok=*/false));
        Pass a2 = ps;
        a2.O = sc2->Os; a2.G = ps.G ? sc2->Gs : nullptr; a2.Mrows = k2;
        auto sliced2 = [&](PackParams& pk) { pk.src = sc2->Rs; pk.R = k2; pk.s_k = 1; pk.s_r = ps.K; pk.conv = 0; };
        sliced2(a2.row.pk);
        if (ps.twin) sliced2(a2.row2.pk);
        a2.cache = ps.col.expanded ? ps.cache : ps.cache ? &sc2->aplane : nullptr;
        a2.ecache = nullptr; a2.scores_keep = SA2; a2.no_select = true; a2.crange = r2; a2.crange_blk = rblk2;
        int hblk_a2[2 * MIR_BLK];                         // (stage A2 runs on the first tier's ranges; the second hull overwrites hblk)
        std::copy(hblk, hblk + 2 * MIR_BLK, hblk_a2);
        a2.host_lo = hlo; a2.host_hi = hhi; a2.host_rblk = hblk_ok ? hblk_a2 : nullptr;
        a2.S1_pre = S1s; a2.S2_pre = S2s; a2.s_ready = true;
        g_stage = 4;
        { const int r_ = run_pass(c, a2); g_stage = 0; if (r_) return r_; }
        if (!c.dry) {
            PruneParams pp2 = pp;                 // same bound L* (stage B1's totals), the tighter partial sums, hull into r3
            pp2.SA = SA2; pp2.r_out = r3; pp2.r_host = hm ? hm + 2 : nullptr; pp2.rblk = rblk3; pp2.rblk_host = hm ? hm + MIR_RB2 : nullptr;
            CHK(enqueue(c, KERN(HullParams, k_prune_hull), dim3(1), dim3(256), 0, HullParams{pp2, hsl}));
            int h[2] = {0, 1};
            if (!hm) CHK(q_d2h(c, h, r3, sizeof h));
            CHK(q_sync(c));
            if (hm) { h[0] = reinterpret_cast<volatile int*>(hm)[2]; h[1] = reinterpret_cast<volatile int*>(hm)[3]; }
            if (tune(TUNE_PRINT) > 0) fprintf(stderr, "[p4v] second tier: %d survivors of the %d-row slice -> %d of the %d-row slice (M %d N %d K %d)\n", nsurv, k, std::max(0, h[1] - h[0]), k2, ps.Mrows, ps.Ncols, ps.K);
            if (h[0] >= h[1]) return select_without_b2();
            rB = r3; rBblk = rblk3;
            hlo = h[0]; hhi = h[1];
            if (hm && ps.nj <= MIR_BLK) { for (int i = 0; i < 2 * ps.nj; ++i) hblk[i] = reinterpret_cast<volatile int*>(hm)[MIR_RB2 + i]; }
            else hblk_ok = false;
        }
    }
    // stage B2: whatever else survives, on all samples (an empty range when stage B1 already covers the survivors)
    Pass b2 = ps;
    b2.crange = rB; b2.crange_blk = rBblk; b2.scores_keep = S2; b2.no_select = true;
    b2.host_lo = hlo; b2.host_hi = hhi; b2.host_rblk = hblk_ok ? hblk : nullptr;
    b2.S1_pre = S1s; b2.S2_pre = S2s; b2.s_ready = true;
    g_stage = 3;
    { const int r_ = run_pass(c, b2); g_stage = 0; if (r_) return r_; }
    // Stage B2's range is the hull of ALL survivors, stage B1's candidates among them: its table decides.  The merge only fills
    // entries stage B2 did NOT evaluate (-inf) with stage B1's -- i.e. it matters when the device-side range was empty and this
    // path ran without the host knowing (no pass memo): then every block's only survivor is stage B1's candidate.
    if (!c.dry) {
        if (virt) CHK(enqueue(c, KERN(MergeVirtParams, k_merge_virtual), dim3(cdiv(ps.nj, 64)), dim3(64), 0, MergeVirtParams{S2, SB, best_idx, ps.nj}));
        else CHK(enqueue(c, KERN(MergeParams, k_merge_scores), dim3(cdiv((long)tab, 256)), dim3(256), 0, MergeParams{S2, SB, (int)tab}));
        CHK(launch_pass_select(c, ps, S2));
    }
    c.ws.off = mark;
    return 0;
}

// ---- split search of the split-of-softmax matmul in one kernel (k_sos_split) -------------------------------------------------
struct SosSplitJob {
    SliceCache* scache; bool host_sync_ok, prunable;      // exact candidate pruning (run_sos_split_pruned)
    SosSplitParams kp;
    int epi; double norm;
    const float* cands; float* split; float* A_iv; float aux_div;
    float* scores_out; int scores_out_ld; int32_t* best_out;
};
bool sos_split_ok(int M, int K, int N, bool cosm) {
    return !cosm && K <= 200 && M <= 256 && N <= 64 && !(g_variant & 131072);
}
template <int KS> int launch_sos_split_ks(Ctx& c, const SosSplitParams& kp, int epi, const StatInfo* si) {
    const dim3 grid(kp.halves, kp.Z), block(256);
    const size_t lds = (size_t)2 * KS * 64 * sizeof(float);
    P4V_EPI4(epi, return enqueue(c, KERN_T(SosSplitParams, k_sos_split, KS, E), grid, block, lds, kp, si))
}
// one launch of the split-search kernel on `kp` (+ k_finish into `scores`, [C] floats); `crange`: device-side candidate range
SelectParams sos_select_params(const SosSplitJob& j, const float* scores) {
    return SelectParams{scores, j.kp.C, 1, j.cands, 1, 0, 0, j.split, 0, 0, j.A_iv, j.aux_div, j.scores_out, j.scores_out_ld, j.best_out};
}
int sos_sweep(Ctx& c, SosSplitJob& j, SosSplitParams kp, const int* crange, float* scores, int known_cands = -1) {
    const int slots = kp.halves * 4;
    float* part = c.ws.get<float>((size_t)kp.C * kp.Z * slots);
    if (!c.ws.ok()) return fail(P4V_ERR_WORKSPACE, "workspace too small: need >= %zu bytes", c.ws.off);
    kp.part = part; kp.crange = crange;
    if (!c.dry) {
        CHK(q_fill(c, part, 0, sizeof(float) * (size_t)kp.C * kp.Z * slots));   // slots of all-padding waves
        const int KS = kp.K <= 64 ? 32 : kp.K <= 144 ? 72 : 100;
        double frac = 1.0;
        if (g_stat_on && crange && known_cands >= 0) frac = (double)known_cands / kp.C;      // (no extra round trip: see run_pass)
        else if (g_stat_on && crange) {
            int h[2] = {0, kp.C};
            CHK(q_d2h(c, h, crange, sizeof h));
            CHK(q_sync(c));
            frac = (double)std::max(0, std::min(h[1], kp.C) - std::max(h[0], 0)) / kp.C;
        }
        const StatInfo si{11, frac * (double)kp.Z * kp.halves * 128 * (2.0 * KS) * 64 * kp.C, frac * (double)kp.Z * kp.M * kp.K * kp.N * kp.C, g_stage, kp.halves, kp.Z,
                          4.0 * ((double)kp.Z * kp.M * kp.K + (double)kp.Z * kp.K * kp.N) + 8.0 * (double)kp.Z * kp.M * kp.N};
        const StatInfo* sp = &si;
        if (KS == 32) CHK(launch_sos_split_ks<32>(c, kp, j.epi, sp));
        else if (KS == 72) CHK(launch_sos_split_ks<72>(c, kp, j.epi, sp));
        else CHK(launch_sos_split_ks<100>(c, kp, j.epi, sp));
    }
    FinishParams fp{part, (long)kp.Z * slots, (long)slots, slots, 1, kp.Z, slots, kp.C, 0, 1, 1, j.norm, scores, crange};
    return launch_finish(c, fp);
}
int sos_select(Ctx& c, SosSplitJob& j, const float* scores) {
    return launch_select(c, sos_select_params(j, scores));
}
int run_sos_split(Ctx& c, SosSplitJob& j) {
    const size_t mark = c.ws.off;
    float* scores = c.ws.get<float>((size_t)j.kp.C);
    CHK(sos_sweep(c, j, j.kp, nullptr, scores));
    CHK(sos_select(c, j, scores));
    c.ws.off = mark;
    return 0;
}
// The split search with the exact pruning of run_pass_pruned: the 20 splits on the 16 heaviest query rows of every (image, head),
// the winner on everything (the bound), whatever survives on everything.  Runs once per module (its result does not depend on
// the intervals: the later rounds are memo hits), before any other pass of the module -- it is what builds the module's slice.
int run_sos_split_pruned_impl(Ctx& c, SosSplitJob& j);
int run_sos_split_pruned(Ctx& c, SosSplitJob& j) {
    if (!(g_variant & 134217728) || c.dry || !j.split) return run_sos_split_pruned_impl(c, j);   // (same scratch as the pruned search)
    std::vector<float> keep;                      // debug cross-check against the full split search (see run_pass_pruned)
    CHK(run_sos_split_pruned_impl(c, j));
    CHK(prune_crosscheck_begin(c, j.split, 1, keep));
    CHK(run_sos_split(c, j));
    return prune_crosscheck_end(c, j.split, 1, keep, "split search");
}
int run_sos_split_pruned_impl(Ctx& c, SosSplitJob& j) {
    const SosSplitParams& kp = j.kp;
    const int k = 16;
    if (!j.prunable || !j.scache || j.scores_out || j.best_out || (g_variant & 4194304) || kp.M < 64 || j.scache->loose) {
        PRUNE_COUNT((!j.prunable || !j.scache || j.scores_out || j.best_out || (g_variant & 4194304)) ? 3 : 2);
        return run_sos_split(c, j);
    }
    SliceGeo geo{false, kp.Z, kp.M, k, kp.N, kp.K, kp.O, (kp.wt_mode == 1 ? kp.G : nullptr), kp.wt_mode, (long)kp.N, kp.A, kp.a_r, kp.a_k,
                 kp.zdiv, kp.a_z2, kp.a_z, false, PackParams{}, k};
    SliceCache* sc = j.scache;
    CHK(slice_alloc(c, sc, geo, true, false));
    const size_t mark = c.ws.off;
    float* SA = c.ws.get<float>((size_t)kp.C);
    float* SB = c.ws.get<float>((size_t)kp.C);
    float* S2 = c.ws.get<float>((size_t)kp.C);
    int* r1 = c.ws.get<int>(4);
    int* r2 = r1 + 2;
    if (!c.ws.ok()) return fail(P4V_ERR_WORKSPACE, "workspace too small: need >= %zu bytes", c.ws.off);
    if (!c.dry) {
        CHK(slice_fill(c, sc, geo, j.host_sync_ok));
        if (sc->loose) { c.ws.off = mark; PRUNE_COUNT(2); return run_sos_split(c, j); }
    }
    PRUNE_COUNT(0);
    SosSplitParams a = kp;                       // stage A: dense slices [Z][16][K] / [Z][16][N]
    a.A = sc->Rs; a.a_k = 1; a.a_r = kp.K; a.a_z = (long)k * kp.K; a.a_z2 = (long)kp.zdiv * k * kp.K;
    a.O = sc->Os; a.G = (kp.wt_mode == 1) ? sc->Gs : sc->Os; a.M = k; a.halves = 1;
    PruneParams pp{SA, SB, kp.C, 1, prune_margin(), r1, r1, 0, nullptr, nullptr, 0, 0, 0, nullptr};
    g_stage = 1;
    { const int r_ = sos_sweep(c, j, a, nullptr, SA); g_stage = 0; if (r_) return r_; }
    CHK(enqueue(c, KERN(PruneParams, k_prune_pick), dim3(1), dim3(256), 0, pp));
    g_stage = 2;
    { const int r_ = sos_sweep(c, j, kp, r1, SB, 1); g_stage = 0; if (r_) return r_; }   // B1 (the slice winner)
    pp.r_out = r2;                                // (+ the selection from its totals when nothing else survives)
    if (!c.dry) {
        SelectParams hsl = sos_select_params(j, SB);
        attach_mirror(hsl);
        CHK(enqueue(c, KERN(HullParams, k_prune_hull), dim3(1), dim3(256), 0, HullParams{pp, hsl}));
    }
    int survivors = -1;
    if (j.host_sync_ok && !c.dry) {
        int h[2] = {0, 1};
        CHK(q_d2h(c, h, r2, sizeof h));
        CHK(q_sync(c));
        if (h[0] >= h[1]) { PRUNE_COUNT(1); c.ws.off = mark; return 0; }
        survivors = std::max(0, std::min(h[1], kp.C) - std::max(h[0], 0));
    }
    g_stage = 3;
    { const int r_ = sos_sweep(c, j, kp, r2, S2, survivors); g_stage = 0; if (r_) return r_; }   // B2
    if (!c.dry) {
        CHK(enqueue(c, KERN(MergeParams, k_merge_scores), dim3(1), dim3(256), 0, MergeParams{S2, SB, kp.C}));
    }
    CHK(sos_select(c, j, S2));
    c.ws.off = mark;
    return 0;
}

// ---- exact memoisation of search passes ---------------------------------------------------------------------
// Every candidate table is built once from the initial interval (reference linear.py:544-545), so a search pass
// is a deterministic function of the counterpart's CURRENT interval only.  Rounds 2-3 often see an interval that
// was already evaluated (the alternation has converged): the pass would recompute, bit for bit, the selection it
// produced before.  Such a pass is skipped and its recorded output restored.  Exact by construction (the kernels
// are deterministic); disabled when the caller asks for score tables, with desc.reserved bit 1, or P4V variant 512.
struct PassMemo {
    struct Entry { std::vector<float> in, out; };
    std::vector<Entry> entries;
    const std::vector<float>* find(const std::vector<float>& in) const {
        for (const auto& e : entries)
            if (e.in.size() == in.size() && std::memcmp(e.in.data(), in.data(), in.size() * sizeof(float)) == 0) return &e.out;
        return nullptr;
    }
};

int read_dev(Ctx& c, const float* d, int n, std::vector<float>& h) {
    h.resize(n);
    if (IvMirror* m = mirror_of(d); m && m->valid && m->count == n) {      // the selection that wrote `d` also wrote the mirror
        CHK(q_sync(c));
        const volatile float* src = m->host;
        for (int i = 0; i < n; ++i) h[i] = src[i];
        return 0;
    }
    CHK(q_d2h(c, h.data(), d, sizeof(float) * n));
    return q_sync(c);
}

int write_dev(Ctx& c, float* d, const std::vector<float>& h) {
    if (IvMirror* m = mirror_of(d)) m->valid = false;
    CHK(q_h2d(c, d, h.data(), sizeof(float) * h.size()));
    if (c.grp) return 0;                    // (a group's queue holds a copy of the data: the restore needs no round trip)
    return q_sync(c);                       // h may go out of scope
}

// binds the interval vectors of one *_impl call (the ones its pass memo reads back) to the stream's mapped block
struct MirrorScope {
    MirrorScope(Ctx& c, bool on, const float* a, const float* b = nullptr, const float* d3 = nullptr) {
        float* base = (on && !c.dry && tune(TUNE_B1_PATH) != 8) ? reinterpret_cast<float*>(host_mirror(c)) : nullptr;
        const float* devs[MIR_SLOTS] = {a, b, d3};
        for (int i = 0; i < MIR_SLOTS; ++i)
            g_mir[i] = IvMirror{(base && devs[i]) ? devs[i] : nullptr, base ? base + MIR_HDR + i * MIR_SLOT : nullptr, false, 0};
    }
    ~MirrorScope() { for (auto& m : g_mir) m = IvMirror{}; }
    MirrorScope(const MirrorScope&) = delete;
    MirrorScope& operator=(const MirrorScope&) = delete;
};

// Which parts of calibration_step2 one call runs.  The fused entry points run everything (ST_ALL: initialisation, then
// search_round x {first operand, second operand} with memoisation); the granular entry points of the C ABI
// (p4v_amax_init_*, p4v_*_search_*) run ONE part with the candidate table supplied by the caller, the way the reference's
// _initialize_intervals / _search_best_*_interval methods are called one by one (linear.py:380-397,455-533).
enum { ST_INIT = 1, ST_S1 = 2, ST_S2 = 4, ST_ALL = 7 };
struct Stage {
    int mask = ST_ALL;
    const float* cands1 = nullptr;   // caller's candidate table of the first operand  (w / A), used when ST_INIT is not set
    const float* cands2 = nullptr;   // ... of the second operand (a / B)
    bool full() const { return mask == ST_ALL; }
    bool searches() const { return (mask & (ST_S1 | ST_S2)) != 0; }
};

PackParams pack2d(const float* src, long rows, long cols, long ld) {
    PackParams p{};
    p.src = src; p.s_z = 0; p.s_r = ld; p.s_k = 1; p.Z = 1; p.R = (int)rows; p.K = (int)cols;
    p.mode = PACK_SYM; p.nblk_r = 1; p.nblk_k = 1;
    return p;
}

// ------------------------------------------------------------------------------------------------
// Linear
// ------------------------------------------------------------------------------------------------
int linear_impl(const p4v_linear_desc* d, const float* W, const float* bias, const float* X, const float* O,
                const float* G, const float* mult, float* w_iv, float* a_iv, float* scores_out, int32_t* best_out,
                Ctx& c, float* fwd_out = nullptr, const Stage& sg = Stage{}) {
    // fwd_out != nullptr: quant_forward (linear.py:62-67 / 601-607) -- w_iv / a_iv are INPUTS, nothing is searched
    const int M = d->batch * d->tokens, K = d->in_features, N = d->out_features;
    const int nV = d->n_V, nH = d->n_H, nA = d->n_a;
    if (M <= 0 || K <= 0 || N <= 0 || nV <= 0 || nH <= 0 || nA <= 0 || d->eq_n <= 0)
        return fail(P4V_ERR_INVALID, "linear: non-positive dimension");
    if (N % nV || K % nH || K % nA) return fail(P4V_ERR_UNSUPPORTED, "linear: n_V/n_H/n_a must divide the layer (reference ignores remainders, linear.py:118)");
    if (d->w_bit > 8 || d->a_bit > 8 || d->w_bit < 2 || d->a_bit < 2) return fail(P4V_ERR_UNSUPPORTED, "linear: bit widths 2..8 supported");
    const int crb_rows = N / nV, crb_cols = K / nH, crb_acts = K / nA;
    const int wq = 1 << (d->w_bit - 1), aq = 1 << (d->a_bit - 1);
    const float a_neg = (float)(0.16997124254703522 / aq);   // linear.py:574
    int epi, wt_mode;
    metric_epi(d->metric, &epi, &wt_mode);
    const bool cosm = epi == EPI_COS;
    if (wt_mode == 1 && !G && !fwd_out && sg.searches()) return fail(P4V_ERR_INVALID, "linear: hessian metric needs raw_grad (linear.py:418)");
    const bool general = (nH > 1 || nA > 1 || (d->reserved & 1) || (cosm && d->twin_postgelu && !fwd_out));
    const bool i8 = !general;
    const bool twin = d->twin_postgelu && i8;
    if (cosm && !fwd_out && (nH > 1 || nA > 1)) return fail(P4V_ERR_UNSUPPORTED, "linear: cosine with n_H>1 / n_a>1 is not implemented on the GPU");
    const int ncand = d->eq_n + 1;

    cvt_bias(c);     // (first call of the process: probe the conversion quant16_sat8 relies on)
    // ---- interval initialisation (linear.py:380-397 / 576-599) ---------------------------------------
    unsigned* enc_w = c.ws.get<unsigned>((size_t)nV * nH);
    unsigned* enc_a = c.ws.get<unsigned>((size_t)nA);
    float* w_cands_ws = c.ws.get<float>((size_t)ncand * nV * nH);
    float* a_cands_ws = c.ws.get<float>((size_t)ncand * nA);
    const float* w_cands = (sg.mask & ST_INIT) ? w_cands_ws : sg.cands1;
    const float* a_cands = (sg.mask & ST_INIT) ? a_cands_ws : sg.cands2;
    float* Ufold = twin ? c.ws.get<float>((size_t)M * N) : nullptr;   // twin activation search: folded target
    float* w_mix = c.ws.get<float>((size_t)ncand * nV * nH);   // general path: candidates of block column h only
    float* a_mix = c.ws.get<float>((size_t)ncand * nA);
    if (!c.ws.ok()) return fail(P4V_ERR_WORKSPACE, "workspace too small");
    // (the weight side first: it depends on no captured tensor -- inside a group that was handed a "capture done" event it runs,
    // with the candidate planes of the weights further down, while the capture passes are still on the GPU)
    if (!fwd_out && (sg.mask & ST_INIT)) {
        const long stw[4] = {0, 0, K, 1};
        CHK(launch_absmax(c, W, stw, 1, 1, N, K, nV, nH, crb_rows, crb_cols, 0, enc_w));
        CHK(launch_interval(c, enc_w, nV * nH, (float)(wq - 0.5), d->init_layerwise, w_iv));
        if (sg.searches()) CHK(launch_cands(c, mult, w_iv, ncand, nV * nH, w_cands_ws));
    }
    auto init_activation_side = [&]() -> int {
        if (!fwd_out && (sg.mask & ST_INIT)) {
            const long stw[4] = {0, 0, K, 1};
            CHK(launch_absmax(c, X, stw, 1, 1, M, K, 1, nA, M, crb_acts, d->twin_postgelu ? 1 : 0, enc_a));
            CHK(launch_interval(c, enc_a, nA, (float)(aq - 0.5), d->init_layerwise, a_iv));
            if (sg.searches()) CHK(launch_cands(c, mult, a_iv, ncand, nA, a_cands_ws));
        }
        return 0;
    };
    if (fwd_out || !sg.full()) {                 // (quant_forward, the granular entry points: no capture to overlap with)
        CHK(q_wait_inputs(c));
        CHK(init_activation_side());
    }
    if (!fwd_out && !sg.searches()) return 0;

    auto x_operand = [&](bool expanded, const float* scales, int sc_cs) {
        Operand op{};
        op.present = true; op.expanded = expanded;
        op.pk = pack2d(X, M, K, K);
        op.pk.scales = scales; op.pk.sc_cs = sc_cs;
        op.pk.lo = -aq; op.pk.hi = aq - 1;
        if (nA > 1) { op.pk.blk_mode = 1; op.pk.blk_div = INT_MAX; op.pk.nblk_r = 1; op.pk.nblk_k = nA; op.pk.blk_div2 = crb_acts; }
        if (d->twin_postgelu) {
            if (i8) { op.pk.lo = 0; }
            else { op.pk.mode = PACK_TWIN_SIM; op.pk.lo = -aq; op.pk.neg_scale = a_neg; }
        }
        return op;
    };
    auto xneg_operand = [&]() {
        Operand op{};
        op.present = true; op.expanded = false;
        op.pk = pack2d(X, M, K, K);
        op.pk.scales = nullptr; op.pk.neg_scale = a_neg; op.pk.lo = -aq; op.pk.hi = 0;
        return op;
    };
    auto w_operand = [&](bool expanded, const float* scales, int sc_cs, bool by_vblock) {
        Operand op{};
        op.present = true; op.expanded = expanded;
        if (by_vblock) {   // swapped cosine sweep: one GEMM per V block
            op.pk = pack2d(W, crb_rows, K, K);
            op.pk.s_z = (long)crb_rows * K;
            op.pk.blk_mode = 2; op.pk.blk_div = nV;
            if (nH > 1) return op;  // unreachable: general path is not swapped per block
        } else {
            op.pk = pack2d(W, N, K, K);
            op.pk.blk_mode = 1; op.pk.blk_div = crb_rows; op.pk.nblk_r = nV; op.pk.nblk_k = nH;
            op.pk.blk_div2 = nH > 1 ? crb_cols : 0;
        }
        op.pk.scales = scales; op.pk.sc_cs = sc_cs;
        op.pk.lo = -wq; op.pk.hi = wq - 1;
        return op;
    };

    if (fwd_out) {
        // out = Q_a(x) . Q_w(W)^T + bias as ONE integer GEMM: grid indices packed to int8 planes (the twin's two ranges
        // as two planes), scales s_a * s_w[block] applied to the int32 accumulators in the epilogue
        Pass fp{};
        fp.i8 = i8; fp.twin = twin; fp.epi = EPI_FWD; fp.wt_mode = 0; fp.eq_n = 1; fp.K = K;
        fp.Z = 1; fp.Mrows = M; fp.Ncols = N;
        fp.row = x_operand(false, a_iv, 0);
        if (twin) fp.row2 = xneg_operand();
        fp.col = w_operand(false, w_iv, 0, false);
        fp.use_s1 = i8; fp.s_cs = nV; fp.sb_mode = 1; fp.sb_div = crb_rows;
        fp.s1 = ScaleParams{a_iv, 0, 0, 0.f, w_iv, 0, 1, 0.f, 0, 0, nullptr};
        fp.s2 = ScaleParams{nullptr, 0, 0, a_neg, w_iv, 0, 1, 0.f, 0, 0, nullptr};
        fp.bias = bias; fp.bias_axis = 0; fp.bias_zs = 0;
        fp.O = fwd_out; fp.G = nullptr; fp.o_ms = N; fp.o_ns = 1; fp.o_inner = INT_MAX;
        fp.nj = 1; fp.store_out = fwd_out;
        return run_pass(c, fp);
    }
    const bool memo_on = sg.full() && !c.dry && !scores_out && !best_out && !(d->reserved & 2) && !(g_variant & 512);
    PassMemo memo_w, memo_a;
    MirrorScope mirrors(c, memo_on, w_iv, a_iv);
    PlaneCache plane_w, plane_a;
    EpiCache epi_w, epi_a;
    SliceCache slice, slice2;
    const bool keep_planes = sg.full() && d->search_round > 1;
    std::vector<float> key, val, host_w, host_a;     // host copies of the current intervals, when a pass just moved them
    bool host_w_ok = false, host_a_ok = false;
    const int n_rounds = sg.full() ? d->search_round : 1;
    auto slot = [&](int round, int which) { return sg.full() ? round * 2 + which : 0; };   // granular call: one table
    // Cosine on k_sweep6 (round 6): the layer as ONE GEMM in the orientation of the difference metrics (rows = samples, columns =
    // features), the register-stationary sweep with the cosine epilogues EPI_COS / EPI_COS_T; the V blocks are runs of 64-feature
    // slabs of k_finish_cos's table.  Otherwise (K > 768, blocks that are not whole slabs, variant 2048): k_sweep2 on the swapped
    // operands, one GEMM per V block.
    const bool cos6 = cosm && i8 && !twin && !general && nH == 1 && nA == 1 && (nV == 1 || crb_rows % 64 == 0) &&
                      sweep6_supported((int)(rup(K, 64) / 64)) && !(g_variant & (4 | 16 | 2048)) && !g_force_v1;
    // ... and on k_sweep7 for K >= 1024 (fc2): 128-feature slabs; the conditions are run_pass's for that kernel
    const bool cos7 = cosm && i8 && !twin && !general && nH == 1 && nA == 1 && (nV == 1 || crb_rows % 128 == 0) && !cos6 &&
                      rup(K, 64) >= 1024 && rup(K, 64) % 256 == 0 && N % 32 == 0 && (long)M * N * 4 < (1L << 32) &&
                      (c.dry || (((unsigned long long)O) & 15) == 0) && !(g_variant & (2048 | 32768)) && !g_force_v1;
    auto cos6_pass = [&](Pass& ps, int j_mode) {
        ps.cos6 = cos6; ps.cos7 = cos7; ps.G = nullptr; ps.wt_mode = 0; ps.prunable = false; ps.scache = nullptr; ps.scache2 = nullptr;
        ps.cos_ZB = 1; ps.cos_ZV = nV; ps.cos_j_mode = j_mode;
        ps.norm = 1.0 / (double)d->tokens;
    };
    // the weight search pass of (round, column block h): reference linear.py:455-495
    auto w_search_pass = [&](int round, int h, const float* wc, int wc_cs, bool memo_w_on) -> Pass {
        Pass ps{};
        ps.i8 = i8; ps.twin = twin; ps.epi = epi; ps.wt_mode = wt_mode; ps.eq_n = d->eq_n; ps.K = K;
        ps.cache = (keep_planes && nH == 1) ? &plane_w : nullptr;   // (n_H > 1: the table mixes in the current interval)
        ps.ecache = keep_planes ? &epi_w : nullptr;
        ps.nj = nV; ps.cands = w_cands; ps.cand_cs = nV * nH; ps.cand_js = nH; ps.cand_off = h;
        ps.interval = w_iv; ps.out_js = nH; ps.out_off = h;
        ps.scores_out = scores_out ? scores_out + ((long)slot(round, 0) * d->eq_n) * nV : nullptr;
        ps.scores_out_ld = nV;
        ps.best_out = best_out ? best_out + (long)slot(round, 0) * nV : nullptr;
        if (h > 0) { ps.scores_out = nullptr; ps.best_out = nullptr; }  // tables of the first column block only
        if (!cosm || cos6 || cos7) {
            ps.Z = 1; ps.Mrows = M; ps.Ncols = N;
            ps.row = x_operand(false, a_iv, 0);
            if (twin) { ps.row2 = xneg_operand(); ps.twin_disjoint = true; }    // linear.py:605-606: clamp(.,0,q-1) / clamp(.,-q,0)
            ps.col = w_operand(true, wc, wc_cs, false);
            ps.use_s1 = i8; ps.s_cs = nV; ps.sb_mode = 1; ps.sb_div = crb_rows;
            ps.s1 = ScaleParams{a_iv, 0, 0, 0.f, w_cands, nV, 1, 0.f, 0, 0, nullptr};
            ps.s2 = ScaleParams{nullptr, 0, 0, a_neg, w_cands, nV, 1, 0.f, 0, 0, nullptr};
            ps.bias = bias; ps.bias_axis = 0; ps.bias_zs = 0;
            ps.O = O; ps.G = G; ps.o_zs = 0; ps.o_bs = 0; ps.o_ms = N; ps.o_ns = 1; ps.o_inner = INT_MAX;
            ps.j_mode = 1; ps.j_div = crb_rows;
            ps.norm = 1.0 / ((double)d->tokens * crb_rows);
            ps.prunable = !(d->reserved & 8); ps.scache = &slice; ps.scache2 = &slice2; ps.host_sync_ok = memo_w_on;
            if (cos6 || cos7) cos6_pass(ps, 1);
        } else {
            // swapped: rows = features of V block z, cols = samples
            ps.Z = nV; ps.Mrows = crb_rows; ps.Ncols = M;
            ps.row = w_operand(true, wc, wc_cs, true);
            ps.col = x_operand(false, a_iv, 0);
            ps.col_zs_shared = 1;
            ps.use_s1 = i8; ps.s_cs = nV; ps.sb_mode = 2; ps.sb_div = nV;
            ps.s1 = ScaleParams{a_iv, 0, 0, 0.f, w_cands, nV, 1, 0.f, 0, 0, nullptr};
            ps.bias = bias; ps.bias_axis = 1; ps.bias_zs = crb_rows;
            ps.O = O; ps.G = nullptr; ps.o_zs = crb_rows; ps.o_bs = 0; ps.o_ms = 1; ps.o_ns = N; ps.o_inner = INT_MAX;
            ps.cos_ZB = 1; ps.cos_ZV = nV; ps.cos_j_mode = 1;
            ps.norm = 1.0 / (double)d->tokens;
        }
        return ps;
    };
    if (sg.full()) {
        // The 100 candidate planes of the weights (8.5 GB per ViT-B calibration over its 48 Linear layers) need the weights and
        // their candidate table, nothing captured: packed here, before the call waits for its captured tensors.
        if (keep_planes && nH == 1 && !cosm && (sg.mask & ST_S1)) {
            Pass pp = w_search_pass(0, 0, w_cands, nV * nH, memo_on);
            pp.pack_only = true;
            CHK(run_pass(c, pp));
        }
        CHK(q_wait_inputs(c));
        CHK(init_activation_side());
    }
    for (int round = 0; round < n_rounds; ++round) {
        // ================= weight search (linear.py:455-495) =================
        // With n_H > 1 the weight search is a coordinate descent over the column blocks: block h is swept with the other
        // blocks at the interval ENTERING the pass (linear.py:468), so its result also depends on w_iv, not only on a_iv
        // (same for n_a > 1 and the activation search).  The memo keys on the counterpart interval alone and is therefore
        // only used where that is the whole input of the pass.
        const bool memo_w_on = memo_on && nH == 1, memo_a_on = memo_on && nA == 1;
        bool skip_w = false;
        if (memo_w_on) {
            if (host_a_ok) key = host_a; else CHK(read_dev(c, a_iv, nA, key));     // (just read back / written by the last pass)
            if (const auto* hit = memo_w.find(key)) { CHK(write_dev(c, w_iv, *hit)); skip_w = true; g_memo_hits++; host_w = *hit; host_w_ok = true; }
        }
        for (int h = 0; h < nH && !skip_w && (sg.mask & ST_S1); ++h) {
            const float* wc = w_cands;
            int wc_cs = nV * nH;
            if (general && nH > 1) {
                // candidates replace column block h only, the others keep the current interval (linear.py:468-469)
                if (!c.dry) {
                    ScaleParams mp{};  // w_mix[c][j] = (j % nH == h) ? w_cands[c][j] : w_iv[j] -- done with two launches
                    mp.x = w_iv; mp.x_cs = 0; mp.x_js = 1; mp.y = nullptr; mp.y_const = 1.0f; mp.C = ncand; mp.nblk = nV * nH; mp.S = w_mix;
                    CHK(launch_scale(c, mp));
                    for (int v = 0; v < nV; ++v)
                        CHK(q_copy2d(c, w_mix + v * nH + h, sizeof(float) * nV * nH, w_cands + v * nH + h, sizeof(float) * nV * nH, sizeof(float), ncand));
                }
                wc = w_mix;
            }
            Pass ps = w_search_pass(round, h, wc, wc_cs, memo_w_on);
            CHK(run_pass_pruned(c, ps));
        }
        if (memo_w_on && !skip_w) { CHK(read_dev(c, w_iv, nV * nH, val)); memo_w.entries.push_back({key, val}); g_memo_misses++; host_w = val; host_w_ok = true; }
        else if (!skip_w) host_w_ok = false;
        // ================= activation search (linear.py:497-533 / 609-642) =================
        bool skip_a = false;
        if (memo_a_on) {
            if (host_w_ok) key = host_w; else CHK(read_dev(c, w_iv, nV * nH, key));
            if (const auto* hit = memo_a.find(key)) { CHK(write_dev(c, a_iv, *hit)); skip_a = true; g_memo_hits++; host_a = *hit; host_a_ok = true; }
        }
        for (int a = 0; a < nA && !skip_a && (sg.mask & ST_S2); ++a) {
            Pass ps{};
            ps.i8 = i8; ps.twin = twin; ps.epi = epi; ps.wt_mode = wt_mode; ps.eq_n = d->eq_n; ps.K = K;
            const float* ac = a_cands;
            if (general && nA > 1) {
                if (!c.dry) {
                    ScaleParams mp{};
                    mp.x = a_iv; mp.x_cs = 0; mp.x_js = 1; mp.y = nullptr; mp.y_const = 1.0f; mp.C = ncand; mp.nblk = nA; mp.S = a_mix;
                    CHK(launch_scale(c, mp));
                    CHK(q_copy2d(c, a_mix + a, sizeof(float) * nA, a_cands + a, sizeof(float) * nA, sizeof(float), ncand));
                }
                ac = a_mix;
            }
            ps.cache = (keep_planes && nA == 1) ? &plane_a : nullptr;
            ps.ecache = keep_planes ? &epi_a : nullptr;
            ps.nj = 1; ps.cands = a_cands; ps.cand_cs = nA; ps.cand_js = 0; ps.cand_off = a;
            ps.interval = a_iv; ps.out_js = 0; ps.out_off = a;
            ps.scores_out = (scores_out && a == 0) ? scores_out + ((long)slot(round, 1) * d->eq_n) * nV : nullptr;
            ps.scores_out_ld = nV;
            ps.best_out = (best_out && a == 0) ? best_out + (long)slot(round, 1) * nV : nullptr;
            if (!cosm || cos6 || cos7) {
                ps.Z = 1; ps.Mrows = M; ps.Ncols = N;
                ps.row = x_operand(true, ac, nA);
                if (twin) ps.row2 = xneg_operand();
                ps.col = w_operand(false, w_iv, 0, false);
                ps.use_s1 = i8; ps.s_cs = nV; ps.sb_mode = 1; ps.sb_div = crb_rows;
                ps.s1 = ScaleParams{a_cands, 1, 0, 0.f, w_iv, 0, 1, 0.f, 0, 0, nullptr};
                ps.s2 = ScaleParams{nullptr, 0, 0, a_neg, w_iv, 0, 1, 0.f, 0, 0, nullptr};
                ps.bias = bias; ps.bias_axis = 0;
                ps.O = O; ps.G = G; ps.o_ms = N; ps.o_ns = 1; ps.o_inner = INT_MAX;
                ps.j_mode = 0;
                ps.norm = 1.0 / ((double)d->tokens * N);
                ps.prunable = !(d->reserved & 8); ps.scache = &slice; ps.scache2 = &slice2; ps.host_sync_ok = memo_a_on;
                if (twin && wt_mode <= 1 && !(g_variant & 64)) {
                    // Twin activation search: the negative-range plane and the weights are candidate-invariant, so
                    // their product is folded into the target once (U = raw_out - bias - s_neg*s_w*(x_neg . W_q))
                    // and the sweep runs on the positive-range plane alone: half the MFMA work of this pass.
                    Pass fp = ps;
                    fp.twin = false; fp.epi = EPI_STORE; fp.wt_mode = 0; fp.eq_n = 1;
                    fp.row = xneg_operand(); fp.row2 = Operand{};
                    fp.s1 = ps.s2;
                    fp.store_out = Ufold; fp.cache = nullptr;
                    fp.scores_out = nullptr; fp.best_out = nullptr;
                    CHK(run_pass(c, fp));
                    ps.twin = false; ps.row2 = Operand{};
                    ps.O = Ufold; ps.bias = nullptr;
                    // Ufold depends on the CURRENT w_interval: it is rebuilt for every activation pass that runs, so the
                    // fragment-order image of k_sweep6's epilogue operands (built from ps.O) must be rebuilt with it
                    ps.ecache = nullptr;
                    slice.o_src = nullptr;          // ... and so are the gathered rows of the target in the sample slice
                    slice2.o_src = nullptr;         // (both tiers: the second one compares the same pointer and would otherwise keep
                                                    // the previous pass's target rows when two tier-2 activation passes follow each other)
                }
                if (cos6 || cos7) cos6_pass(ps, 0);
            } else {
                ps.Z = nV; ps.Mrows = crb_rows; ps.Ncols = M;
                ps.row = w_operand(false, w_iv, 0, true);
                ps.col = x_operand(true, ac, nA);
                ps.col_zs_shared = 1;
                ps.use_s1 = i8; ps.s_cs = nV; ps.sb_mode = 2; ps.sb_div = nV;
                ps.s1 = ScaleParams{a_cands, 1, 0, 0.f, w_iv, 0, 1, 0.f, 0, 0, nullptr};
                ps.bias = bias; ps.bias_axis = 1; ps.bias_zs = crb_rows;
                ps.O = O; ps.o_zs = crb_rows; ps.o_ms = 1; ps.o_ns = N; ps.o_inner = INT_MAX;
                ps.cos_ZB = 1; ps.cos_ZV = nV; ps.cos_j_mode = 0;
                ps.norm = 1.0 / (double)d->tokens;
            }
            CHK(run_pass_pruned(c, ps));
        }
        if (memo_a_on && !skip_a) { CHK(read_dev(c, a_iv, nA, val)); memo_a.entries.push_back({key, val}); g_memo_misses++; host_a = val; host_a_ok = true; }
        else if (!skip_a) host_a_ok = false;
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------
// MatMul
// ------------------------------------------------------------------------------------------------
int matmul_impl(const p4v_matmul_desc* d, const float* A, const float* B, const float* O, const float* G,
                const float* mult, float* A_iv, float* B_iv, float* split, float* scores_out, int32_t* best_out, Ctx& c,
                float* fwd_out = nullptr, const Stage& sg = Stage{}) {
    // fwd_out != nullptr: quant_forward (matmul.py:140-145; sos: matmul.py:595-598) -- intervals / split are INPUTS
    const int H = d->heads, Z = d->batch * d->heads, M = d->M, K = d->K, N = d->N;
    if (Z <= 0 || M <= 0 || K <= 0 || N <= 0 || d->eq_n <= 0) return fail(P4V_ERR_INVALID, "matmul: non-positive dimension");
    if (d->A_bit > 8 || d->B_bit > 8) return fail(P4V_ERR_UNSUPPORTED, "matmul: bit widths <= 8 supported");
    const int Aq = 1 << (d->A_bit - 1), Bq = 1 << (d->B_bit - 1);
    int epi, wt_mode;
    metric_epi(d->metric, &epi, &wt_mode);
    const bool cosm = epi == EPI_COS;
    if (wt_mode == 1 && !G && !fwd_out && sg.searches()) return fail(P4V_ERR_INVALID, "matmul: hessian metric needs raw_grad");
    if (d->sos && !split) return fail(P4V_ERR_INVALID, "matmul: sos needs d_split");
    cvt_bias(c);     // (first call of the process: probe the conversion quant16_sat8 relies on)
    CHK(q_wait_inputs(c));     // (inside a group with a "capture done" event: everything below reads captured tensors)
    const int ncand = d->eq_n + 1;
    const int NSPLIT = 20;  // matmul.py:636

    unsigned* enc_A = c.ws.get<unsigned>(H);
    unsigned* enc_B = c.ws.get<unsigned>(H);
    float* A_cands_ws = c.ws.get<float>((size_t)ncand * H);
    float* B_cands_ws = c.ws.get<float>((size_t)ncand * H);
    const float* A_cands = (sg.mask & ST_INIT) ? A_cands_ws : sg.cands1;
    const float* B_cands = (sg.mask & ST_INIT) ? B_cands_ws : sg.cands2;
    float* split_cands = c.ws.get<float>(NSPLIT);
    float* A_headwise = c.ws.get<float>(H);   // sos: the inherited head-wise A interval is computed then overwritten (matmul.py:419-440)
    if (!c.ws.ok()) return fail(P4V_ERR_WORKSPACE, "workspace too small");
    if (!fwd_out && (sg.mask & ST_INIT)) {
        const long sa[4] = {d->a_stride[0], d->a_stride[1], d->a_stride[2], d->a_stride[3]};
        const long sb[4] = {d->b_stride[0], d->b_stride[1], d->b_stride[2], d->b_stride[3]};
        float* a_dst = d->sos ? A_headwise : A_iv;
        CHK(launch_absmax(c, A, sa, d->batch, H, M, K, 1, 1, M, K, 0, enc_A));
        CHK(launch_interval(c, enc_A, H, (float)(Aq - 0.5), d->init_layerwise, a_dst));
        CHK(launch_absmax(c, B, sb, d->batch, H, K, N, 1, 1, K, N, 0, enc_B));
        CHK(launch_interval(c, enc_B, H, (float)(Bq - 0.5), d->init_layerwise, B_iv));
        if (sg.searches()) {
            if (!d->sos) CHK(launch_cands(c, mult, A_iv, ncand, H, A_cands_ws));
            CHK(launch_cands(c, mult, B_iv, ncand, H, B_cands_ws));
        }
    }
    if (!fwd_out && !sg.searches()) return 0;
    if (!fwd_out && (sg.mask & ST_S1)) {
        if (d->sos && !c.dry) {
            float sc[NSPLIT];
            for (int i = 0; i < NSPLIT; ++i) sc[i] = (float)std::ldexp(1.0, -i);  // 2**(-i), exact in fp32
            CHK(q_h2d(c, split_cands, sc, sizeof sc));
            if (!c.grp) CHK(q_sync(c));          // sc lives on this stack frame (a group's queue holds a copy)
        }
    }
    // logical [Z][rows][K] views: A rows = m, B rows = n (B given as [K][N])
    auto A_operand = [&](bool expanded, const float* scales, int sc_cs, int mode) {
        Operand op{};
        op.present = true; op.expanded = expanded;
        op.pk = PackParams{};
        op.pk.src = A; op.pk.s_z = d->a_stride[1]; op.pk.s_r = d->a_stride[2]; op.pk.s_k = d->a_stride[3];
        op.pk.s_z2 = d->a_stride[0]; op.pk.zdiv = H;
        op.pk.Z = Z; op.pk.R = M; op.pk.K = K; op.pk.nblk_r = 1; op.pk.nblk_k = 1;
        op.pk.mode = mode; op.pk.scales = scales; op.pk.sc_cs = sc_cs;
        op.pk.blk_mode = (mode == PACK_SYM) ? 2 : 0; op.pk.blk_div = H;
        op.pk.lo = -Aq; op.pk.hi = Aq - 1; op.pk.qm1 = (float)(Aq - 1);
        return op;
    };
    auto B_operand = [&](bool expanded, const float* scales, int sc_cs, int mode) {
        Operand op{};
        op.present = true; op.expanded = expanded;
        op.pk = PackParams{};
        op.pk.src = B; op.pk.s_z = d->b_stride[1]; op.pk.s_r = d->b_stride[3]; op.pk.s_k = d->b_stride[2];
        op.pk.s_z2 = d->b_stride[0]; op.pk.zdiv = H;
        op.pk.Z = Z; op.pk.R = N; op.pk.K = K; op.pk.nblk_r = 1; op.pk.nblk_k = 1;
        op.pk.mode = mode; op.pk.scales = scales; op.pk.sc_cs = sc_cs;
        op.pk.blk_mode = (mode == PACK_SYM) ? 2 : 0; op.pk.blk_div = H;
        op.pk.lo = -Bq; op.pk.hi = Bq - 1;
        return op;
    };
    auto common = [&](Pass& ps) {
        ps.Z = Z; ps.K = K;
        ps.O = O; ps.G = G; ps.wt_mode = wt_mode; ps.epi = epi;
        ps.o_inner = INT_MAX;
        if (!cosm) { ps.Mrows = M; ps.Ncols = N; ps.o_zs = (long)M * N; ps.o_ms = N; ps.o_ns = 1; }
        else { ps.Mrows = N; ps.Ncols = M; ps.o_zs = (long)M * N; ps.o_ms = 1; ps.o_ns = N; ps.G = nullptr;
               ps.cos_ZB = Z; ps.cos_ZV = 1; }
    };

    if (fwd_out) {
        Pass fp{};
        fp.Z = Z; fp.K = K; fp.Mrows = M; fp.Ncols = N;
        fp.i8 = true; fp.twin = d->sos; fp.epi = EPI_FWD; fp.wt_mode = 0; fp.eq_n = 1;
        fp.row = d->sos ? A_operand(false, split, 0, PACK_SOS_HI) : A_operand(false, A_iv, 0, PACK_SYM);
        if (d->sos) fp.row2 = A_operand(false, split, 0, PACK_SOS_LO);
        fp.col = B_operand(false, B_iv, 0, PACK_SYM);
        fp.use_s1 = true; fp.s_cs = H; fp.sb_mode = 2; fp.sb_div = H;
        if (!d->sos) fp.s1 = ScaleParams{A_iv, 0, 1, 0.f, B_iv, 0, 1, 0.f, 0, 0, nullptr};
        else {
            fp.s1 = ScaleParams{nullptr, 0, 0, 1.0f / (float)(Aq - 1), B_iv, 0, 1, 0.f, 0, 0, nullptr};
            fp.s2 = ScaleParams{A_iv, 0, 0, 0.f, B_iv, 0, 1, 0.f, 0, 0, nullptr};
        }
        fp.O = fwd_out; fp.G = nullptr; fp.o_zs = (long)M * N; fp.o_ms = N; fp.o_ns = 1; fp.o_inner = INT_MAX;
        fp.nj = 1; fp.store_out = fwd_out;
        return run_pass(c, fp);
    }
    const bool memo_on = sg.full() && !c.dry && !scores_out && !best_out && !(d->reserved & 2) && !(g_variant & 512);
    PassMemo memo_A, memo_B;
    MirrorScope mirrors(c, memo_on, A_iv, B_iv, d->sos ? split : nullptr);
    PlaneCache plane_A, plane_B;
    SliceCache slice;
    const bool keep_planes = sg.full() && d->search_round > 1;
    std::vector<float> key, val;
    const int nAiv = d->sos ? 1 : H;
    const int n_rounds = sg.full() ? d->search_round : 1;
    for (int round = 0; round < n_rounds; ++round) {
        float* so = scores_out ? scores_out + ((long)(round * 2) * d->eq_n) * H : nullptr;
        int32_t* bo = best_out ? best_out + (long)(round * 2) * H : nullptr;
        // granular call: the one table of this call starts at offset 0 (the B pass below adds eq_n*H / H otherwise)
        float* so_B = so ? (sg.full() ? so + (long)d->eq_n * H : so) : nullptr;
        int32_t* bo_B = bo ? (sg.full() ? bo + H : bo) : nullptr;
        // the A search (or, with sos, the split search against the RAW B: a function of nothing -> always a hit after round 1)
        bool skip_A = false;
        if (memo_on) {
            if (d->sos) key.assign(1, 0.0f); else CHK(read_dev(c, B_iv, H, key));
            if (const auto* hit = memo_A.find(key)) {
                if (d->sos) { std::vector<float> sp(1, (*hit)[0]), ai(1, (*hit)[1]); CHK(write_dev(c, split, sp)); CHK(write_dev(c, A_iv, ai)); }
                else CHK(write_dev(c, A_iv, *hit));
                skip_A = true; g_memo_hits++;
            }
        }
        if (skip_A || !(sg.mask & ST_S1)) {
        } else if (!d->sos) {
            // ---- A search, B fixed at its current head-wise interval (matmul.py:483-522) ----
            Pass ps{};
            common(ps);
            ps.i8 = true; ps.twin = false; ps.eq_n = d->eq_n;
            Operand a = A_operand(true, A_cands, H, PACK_SYM), b = B_operand(false, B_iv, 0, PACK_SYM);
            if (!cosm) { ps.row = a; ps.col = b; } else { ps.row = b; ps.col = a; }
            ps.use_s1 = true; ps.s_cs = H; ps.sb_mode = 2; ps.sb_div = H;
            ps.s1 = ScaleParams{A_cands, H, 1, 0.f, B_iv, 0, 1, 0.f, 0, 0, nullptr};
            ps.j_mode = 2; ps.j_div = H; ps.nj = H; ps.cos_j_mode = 2; ps.cos_j_div = H;
            ps.norm = cosm ? 1.0 / (double)M : 1.0 / ((double)M * N);
            ps.cands = A_cands; ps.cand_cs = H; ps.cand_js = 1; ps.interval = A_iv; ps.out_js = 1;
            ps.cache = keep_planes ? &plane_A : nullptr;
            ps.scores_out = so; ps.scores_out_ld = H; ps.best_out = bo;
            ps.prunable = !cosm && !(d->reserved & 8); ps.scache = &slice; ps.host_sync_ok = memo_on;
            CHK(run_pass_pruned(c, ps));
        } else if (sos_split_ok(M, K, N, cosm)) {
            // ---- split search against the UNQUANTISED B (matmul.py:600-631): A quantised in registers, one kernel ----
            SosSplitJob j{};
            j.kp.A = A; j.kp.a_z2 = d->a_stride[0]; j.kp.a_z = d->a_stride[1]; j.kp.a_r = d->a_stride[2]; j.kp.a_k = d->a_stride[3];
            j.kp.zdiv = H;
            j.kp.B = B; j.kp.b_z2 = d->b_stride[0]; j.kp.b_z = d->b_stride[1]; j.kp.b_k = d->b_stride[2]; j.kp.b_n = d->b_stride[3];
            j.kp.O = O; j.kp.G = G ? G : O;
            j.kp.Z = Z; j.kp.M = M; j.kp.K = K; j.kp.N = N; j.kp.wt_mode = wt_mode; j.kp.C = NSPLIT;
            j.kp.splits = split_cands;
            j.kp.qm1 = (float)(Aq - 1); j.kp.c_inv = 1.0f / (float)(Aq - 1);
            j.kp.lo_top = std::min(std::nearbyintf(1.0f / j.kp.c_inv), (float)(Aq - 1));
            j.kp.halves = cdiv(M, 128);
            j.epi = epi; j.norm = 1.0 / ((double)H * M * N);
            j.cands = split_cands; j.split = split; j.A_iv = A_iv; j.aux_div = (float)(Aq - 1);   // A_interval = split/(qmax-1) (matmul.py:629)
            j.scores_out = (d->eq_n >= NSPLIT) ? so : nullptr; j.scores_out_ld = H; j.best_out = bo;
            j.scache = &slice; j.host_sync_ok = memo_on; j.prunable = !(d->reserved & 8);
            CHK(run_sos_split_pruned(c, j));
        } else {
            // ---- split search against the UNQUANTISED B (matmul.py:600-631): fp32 operands ----
            Pass ps{};
            common(ps);
            ps.i8 = false; ps.twin = false; ps.eq_n = NSPLIT;
            Operand a = A_operand(true, split_cands, 1, PACK_SOS_SIM), b = B_operand(false, nullptr, 0, PACK_RAW);
            if (!cosm) { ps.row = a; ps.col = b; } else { ps.row = b; ps.col = a; }
            ps.use_s1 = false; ps.s_cs = 1; ps.sb_mode = 0;
            ps.j_mode = 0; ps.nj = 1; ps.cos_j_mode = 0;
            ps.norm = cosm ? 1.0 / ((double)H * M) : 1.0 / ((double)H * M * N);
            ps.cands = split_cands; ps.cand_cs = 1; ps.cand_js = 0; ps.interval = split; ps.out_js = 0;
            ps.aux_out = A_iv; ps.aux_div = (float)(Aq - 1);   // A_interval = split/(qmax-1) (matmul.py:629)
            ps.scores_out = (d->eq_n >= NSPLIT) ? so : nullptr; ps.scores_out_ld = H; ps.best_out = bo;
            CHK(run_pass(c, ps));
        }
        if (memo_on && !skip_A) {   // (memo_on implies a full run)
            if (d->sos) { std::vector<float> sp, ai; CHK(read_dev(c, split, 1, sp)); CHK(read_dev(c, A_iv, 1, ai)); val = {sp[0], ai[0]}; }
            else CHK(read_dev(c, A_iv, nAiv, val));
            memo_A.entries.push_back({key, val}); g_memo_misses++;
        }
        bool skip_B = false;
        if (memo_on) {
            CHK(read_dev(c, A_iv, nAiv, key));
            if (const auto* hit = memo_B.find(key)) { CHK(write_dev(c, B_iv, *hit)); skip_B = true; g_memo_hits++; }
        }
        if (!skip_B && (sg.mask & ST_S2)) {
            // ---- B search, A fixed (matmul.py:524-563); with sos, A is the two-range twin (matmul.py:595-598) ----
            Pass ps{};
            common(ps);
            ps.eq_n = d->eq_n;
            const bool twin_rows = d->sos && !cosm;
            ps.i8 = !(d->sos && cosm);   // cosine + sos: fp32 operands (the twin sits on the column side when swapped)
            ps.twin = twin_rows;
            Operand b = B_operand(true, B_cands, H, PACK_SYM);
            Operand a = d->sos ? A_operand(false, split, 0, ps.i8 ? PACK_SOS_HI : PACK_SOS_SIM) : A_operand(false, A_iv, 0, PACK_SYM);
            if (!cosm) { ps.row = a; ps.col = b; if (twin_rows) ps.row2 = A_operand(false, split, 0, PACK_SOS_LO); }
            else { ps.row = b; ps.col = a; }
            ps.use_s1 = ps.i8; ps.s_cs = H; ps.sb_mode = 2; ps.sb_div = H;
            if (!d->sos) ps.s1 = ScaleParams{A_iv, 0, 1, 0.f, B_cands, H, 1, 0.f, 0, 0, nullptr};
            else {
                // high range: k_hi/(q-1) ; low range: k_lo * (split/(q-1)) = k_lo * A_interval
                ps.s1 = ScaleParams{nullptr, 0, 0, 1.0f / (float)(Aq - 1), B_cands, H, 1, 0.f, 0, 0, nullptr};
                ps.s2 = ScaleParams{A_iv, 0, 0, 0.f, B_cands, H, 1, 0.f, 0, 0, nullptr};
            }
            ps.j_mode = 2; ps.j_div = H; ps.nj = H; ps.cos_j_mode = 2; ps.cos_j_div = H;
            ps.norm = cosm ? 1.0 / (double)M : 1.0 / ((double)M * N);
            ps.cands = B_cands; ps.cand_cs = H; ps.cand_js = 1; ps.interval = B_iv; ps.out_js = 1;
            ps.cache = keep_planes ? &plane_B : nullptr;
            ps.scores_out = so_B; ps.scores_out_ld = H;
            ps.best_out = bo_B;
            ps.prunable = !cosm && ps.i8 && !(d->reserved & 8); ps.scache = &slice; ps.host_sync_ok = memo_on;
            CHK(run_pass_pruned(c, ps));
            if (memo_on) { CHK(read_dev(c, B_iv, H, val)); memo_B.entries.push_back({key, val}); g_memo_misses++; }
        }
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Conv2d (patch embedding): fp32 operands (the input stays unquantised with a_bit >= 32, conv.py:544)
// ------------------------------------------------------------------------------------------------
int conv_impl(const p4v_conv_desc* d, const float* W, const float* bias, const float* X, const float* O, const float* G,
              const float* mult, float* w_iv, float* a_iv, float* scores_out, int32_t* best_out, Ctx& c,
              const Stage& sg = Stage{}) {
    const int b = d->batch, ic = d->in_channels, H = d->height, Wd = d->width, oc = d->out_channels;
    const int kh = d->kernel_h, kw = d->kernel_w;
    const int fh = (H + 2 * d->pad_h - d->dil_h * (kh - 1) - 1) / d->stride_h + 1;
    const int fw = (Wd + 2 * d->pad_w - d->dil_w * (kw - 1) - 1) / d->stride_w + 1;
    if (b <= 0 || ic <= 0 || oc <= 0 || fh <= 0 || fw <= 0 || d->eq_n <= 0) return fail(P4V_ERR_INVALID, "conv: bad geometry");
    if (d->w_bit > 8) return fail(P4V_ERR_UNSUPPORTED, "conv: w_bit <= 8 supported");
    const int L = fh * fw, K = ic * kh * kw, M = b * L;
    const int wq = 1 << (d->w_bit - 1);
    const bool aquant = d->a_bit < 32;
    const int aq = aquant ? (1 << (d->a_bit - 1)) : 0;
    if (aquant && !d->channelwise) return fail(P4V_ERR_UNSUPPORTED, "conv: the layer-wise class cannot search activations (reference conv.py:420 raises IndexError); use a_bit=32");
    cvt_bias(c);     // (first call of the process: probe the conversion quant16_sat8 relies on)
    CHK(q_wait_inputs(c));     // (inside a group with a "capture done" event: everything below reads captured tensors)
    int epi, wt_mode;
    metric_epi(d->metric, &epi, &wt_mode);
    const bool cosm = epi == EPI_COS;
    if (wt_mode == 1 && !G && sg.searches()) return fail(P4V_ERR_INVALID, "conv: hessian metric needs raw_grad");
    if (cosm && aquant) return fail(P4V_ERR_UNSUPPORTED, "conv: cosine does not support the activation search (reference conv.py:505-506)");
    const int nw = d->channelwise ? oc : 1;
    const int ncand = d->eq_n + 1;

    unsigned* enc_w = c.ws.get<unsigned>(nw);
    unsigned* enc_a = c.ws.get<unsigned>(1);
    float* w_cands_ws = c.ws.get<float>((size_t)ncand * nw);
    float* a_cands_ws = c.ws.get<float>(ncand);
    const float* w_cands = (sg.mask & ST_INIT) ? w_cands_ws : sg.cands1;
    const float* a_cands = (sg.mask & ST_INIT) ? a_cands_ws : sg.cands2;
    if (!c.ws.ok()) return fail(P4V_ERR_WORKSPACE, "workspace too small");
    if (sg.mask & ST_INIT) {
        const long stw[4] = {0, 0, K, 1};
        CHK(launch_absmax(c, W, stw, 1, 1, oc, K, nw, 1, d->channelwise ? 1 : oc, K, 0, enc_w));
        CHK(launch_interval(c, enc_w, nw, (float)(wq - 0.5), d->init_layerwise, w_iv));
        const long stx[4] = {0, 0, (long)ic * H * Wd, 1};
        CHK(launch_absmax(c, X, stx, 1, 1, b, ic * H * Wd, 1, 1, b, ic * H * Wd, 0, enc_a));
        CHK(launch_interval(c, enc_a, 1, aquant ? (float)(aq - 0.5) : (float)(std::ldexp(1.0, d->a_bit - 1) - 0.5), 0, a_iv));
        if (sg.searches()) {
            CHK(launch_cands(c, mult, w_iv, ncand, nw, w_cands_ws));
            CHK(launch_cands(c, mult, a_iv, ncand, 1, a_cands_ws));
        }
    }
    if (!sg.searches()) return 0;
    // The difference metrics read raw_out / raw_grad as rows of the im2col GEMM, [b * L][oc]: one transposition of the [b][oc][L]
    // tensors (same elements, same sums) gives the sweeps a dense epilogue operand and lets the weight search be pruned like a
    // Linear's (run_pass_pruned: slices are rows).
    const bool prunable = !cosm && sg.full() && !scores_out && !best_out && !(d->reserved & 8) && d->eq_n >= 32;
    float* Ot = prunable ? c.ws.get<float>((size_t)M * oc) : nullptr;
    float* Gt = (prunable && G) ? c.ws.get<float>((size_t)M * oc) : nullptr;
    if (!c.ws.ok()) return fail(P4V_ERR_WORKSPACE, "workspace too small");
    if (prunable && !c.dry) {
        const dim3 grid(cdiv(L, 32), cdiv(oc, 32), b);
        CHK(enqueue(c, KERN(NchwRowsParams, k_nchw_to_rows), grid, dim3(256), 0, NchwRowsParams{O, oc, L, Ot}));
        if (G) CHK(enqueue(c, KERN(NchwRowsParams, k_nchw_to_rows), grid, dim3(256), 0, NchwRowsParams{G, oc, L, Gt}));
    }
    // x as im2col rows.  per_image: Z = batch, rows = pixels of one image (channel-wise cosine reduces over pixels)
    auto x_operand = [&](bool expanded, const float* scales, int sc_cs, bool per_image) {
        Operand op{};
        op.present = true; op.expanded = expanded;
        op.pk = PackParams{};
        op.pk.src = X; op.pk.conv = 1; op.pk.ic = ic; op.pk.H = H; op.pk.W = Wd; op.pk.kh = kh; op.pk.kw = kw;
        op.pk.sh = d->stride_h; op.pk.sw = d->stride_w; op.pk.ph = d->pad_h; op.pk.pw = d->pad_w; op.pk.dh = d->dil_h; op.pk.dw = d->dil_w;
        op.pk.fw = fw; op.pk.L = L;
        op.pk.Z = per_image ? b : 1; op.pk.s_z = per_image ? (long)ic * H * Wd : 0;
        op.pk.R = per_image ? L : M; op.pk.K = K; op.pk.nblk_r = 1; op.pk.nblk_k = 1;
        op.pk.mode = aquant ? PACK_SYM : PACK_RAW; op.pk.scales = aquant ? scales : nullptr; op.pk.sc_cs = sc_cs;
        op.pk.lo = -aq; op.pk.hi = aq - 1;
        return op;
    };
    auto w_operand = [&](bool expanded, const float* scales, int sc_cs) {
        Operand op{};
        op.present = true; op.expanded = expanded;
        op.pk = pack2d(W, oc, K, K);
        op.pk.scales = scales; op.pk.sc_cs = sc_cs;
        op.pk.blk_mode = d->channelwise ? 1 : 0; op.pk.blk_div = 1; op.pk.nblk_r = nw; op.pk.nblk_k = 1;
        op.pk.lo = -wq; op.pk.hi = wq - 1;
        return op;
    };
    auto setup = [&](Pass& ps, bool w_search) {
        ps.i8 = false; ps.twin = false; ps.epi = epi; ps.wt_mode = wt_mode; ps.eq_n = d->eq_n; ps.K = K;
        ps.use_s1 = false; ps.s_cs = 1; ps.sb_mode = 0;
        ps.O = O; ps.G = G; ps.bias = bias;
        Operand xo = x_operand(!w_search, w_search ? a_iv : a_cands, w_search ? 0 : 1, cosm && d->channelwise);
        Operand wo = w_operand(w_search, w_search ? w_cands : w_iv, w_search ? nw : 0);
        if (!cosm) {
            // rows = (image, pixel), cols = oc;  out[b][oc][l]
            ps.Z = 1; ps.Mrows = M; ps.Ncols = oc; ps.row = xo; ps.col = wo;
            ps.o_inner = L; ps.o_bs = (long)oc * L; ps.o_ms = 1; ps.o_ns = L; ps.bias_axis = 0;
            if (prunable) { ps.O = Ot; ps.G = G ? Gt : nullptr; ps.o_inner = INT_MAX; ps.o_bs = 0; ps.o_ms = oc; ps.o_ns = 1; }
        } else if (d->channelwise) {
            // cosine over the pixels of one image per (image, oc) (conv.py:504-508): rows = pixels, z = image
            ps.Z = b; ps.Mrows = L; ps.Ncols = oc; ps.row = xo; ps.col = wo; ps.col_zs_shared = 1;
            ps.o_zs = (long)oc * L; ps.o_inner = INT_MAX; ps.o_ms = 1; ps.o_ns = L; ps.bias_axis = 0; ps.G = nullptr;
            ps.cos_ZB = b; ps.cos_ZV = 1;
        } else {
            // cosine over oc per pixel (conv.py:387): swapped, rows = oc, cols = (image, pixel)
            ps.Z = 1; ps.Mrows = oc; ps.Ncols = M; ps.row = wo; ps.col = xo;
            // element (row=oc, col=m): idx = (m / L)*oc*L + (m % L) + oc_idx*L  -> column index drives the image split
            ps.o_inner = INT_MAX; ps.o_ms = L; ps.o_ninner = L; ps.o_nbs = (long)oc * L; ps.o_ns = 1;
            ps.bias_axis = 1; ps.G = nullptr;
            ps.cos_ZB = 1; ps.cos_ZV = 1;
        }
    };

    const bool memo_on = sg.full() && !c.dry && !scores_out && !best_out && !(d->reserved & 2) && !(g_variant & 512);
    PassMemo memo_w, memo_a;
    MirrorScope mirrors(c, memo_on, w_iv, a_iv);
    PlaneCache plane_w, plane_a;
    SliceCache slice;
    const bool keep_planes = sg.full() && d->search_round > 1;
    std::vector<float> key, val;
    const int n_rounds = sg.full() ? d->search_round : 1;
    for (int round = 0; round < n_rounds; ++round) {
        float* so = scores_out ? scores_out + ((long)(round * 2) * d->eq_n) * nw : nullptr;
        int32_t* bo = best_out ? best_out + (long)(round * 2) * nw : nullptr;
        float* so_a = so ? (sg.full() ? so + (long)d->eq_n * nw : so) : nullptr;
        int32_t* bo_a = bo ? (sg.full() ? bo + nw : bo) : nullptr;
        bool skip_w = false;
        if (memo_on) {
            // with a_bit >= 32 the input is never quantised: the weight search depends on nothing (conv.py:544)
            if (aquant) CHK(read_dev(c, a_iv, 1, key)); else key.assign(1, 0.0f);
            if (const auto* hit = memo_w.find(key)) { CHK(write_dev(c, w_iv, *hit)); skip_w = true; g_memo_hits++; }
        }
        if (!skip_w && (sg.mask & ST_S1)) {   // ---- weight search (conv.py:526-557 / 365-396) ----
            Pass ps{};
            setup(ps, true);
            ps.nj = nw;
            if (!cosm) { ps.j_mode = d->channelwise ? 3 : 0; ps.norm = d->channelwise ? 1.0 / (double)L : 1.0 / ((double)L * oc); }
            else if (d->channelwise) { ps.cos_j_mode = 3; ps.norm = 1.0; }
            else { ps.cos_j_mode = 0; ps.norm = 1.0 / (double)L; }
            ps.cands = w_cands; ps.cand_cs = nw; ps.cand_js = 1; ps.interval = w_iv; ps.out_js = 1;
            ps.cache = keep_planes ? &plane_w : nullptr;
            ps.scores_out = so; ps.scores_out_ld = nw; ps.best_out = bo;
            ps.prunable = ps.prunable_f32 = prunable; ps.scache = &slice; ps.host_sync_ok = memo_on;
            CHK(run_pass_pruned(c, ps));
            if (memo_on) { CHK(read_dev(c, w_iv, nw, val)); memo_w.entries.push_back({key, val}); g_memo_misses++; }
        }
        bool skip_a = false;
        if (memo_on && aquant) {
            CHK(read_dev(c, w_iv, nw, key));
            if (const auto* hit = memo_a.find(key)) { CHK(write_dev(c, a_iv, *hit)); skip_a = true; g_memo_hits++; }
        }
        if (aquant && !skip_a && (sg.mask & ST_S2)) {  // ---- activation search (conv.py:559-589), channel-wise class only ----
            Pass ps{};
            setup(ps, false);
            ps.nj = 1; ps.j_mode = 0; ps.norm = 1.0 / ((double)L * oc);
            ps.cands = a_cands; ps.cand_cs = 1; ps.cand_js = 0; ps.interval = a_iv; ps.out_js = 0;
            ps.cache = keep_planes ? &plane_a : nullptr;
            ps.scores_out = so_a; ps.scores_out_ld = nw;
            ps.best_out = bo_a;
            CHK(run_pass(c, ps));
            if (memo_on) { CHK(read_dev(c, a_iv, 1, val)); memo_a.entries.push_back({key, val}); g_memo_misses++; }
        }
    }
    return 0;
}

// ---- p4v_calibrate_group: the members' calibration_step2 in lock step, launches grouped ----------------------------------------
// Worker threads are persistent (a ViT-B calibration runs 74 members; thread creation per call would cost as much as a search
// pass): a task is handed to an idle worker or a new one is started -- every member of a group must be running at the same time,
// they rendezvous.  The pool object is never destroyed (workers are detached and outlive static destruction order).
struct WorkerPool {
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::function<void()>> tasks;
    int idle = 0;
    void run(std::function<void()> f) {
        std::unique_lock<std::mutex> lk(mu);
        tasks.push_back(std::move(f));
        if (idle >= (int)tasks.size()) { cv.notify_one(); return; }
        lk.unlock();
        std::thread([this] { loop(); }).detach();
    }
    void loop() {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            while (tasks.empty()) { ++idle; cv.wait(lk); --idle; }
            auto f = std::move(tasks.front());
            tasks.pop_front();
            lk.unlock();
            f();
            lk.lock();
        }
    }
};
WorkerPool& worker_pool() { static WorkerPool* p = new WorkerPool; return *p; }

int* group_mirror(hipStream_t st, int dev, int slot) {
    static std::mutex mu;
    static std::unordered_map<std::string, int*> tab;
    char key[64];
    snprintf(key, sizeof key, "%p/%d/%d", (void*)st, dev, slot);
    std::lock_guard<std::mutex> lk(mu);
    auto it = tab.find(key);
    if (it != tab.end()) return it->second;
    int* p = mirror_alloc();
    tab[key] = p;
    return p;
}

int run_group_member(const p4v_group_job& j, Ctx& c) {
    switch (j.kind) {
        case P4V_JOB_LINEAR: {
            const p4v_linear_desc* d = (const p4v_linear_desc*)j.desc;
            if (!j.in[0] || !j.in[2] || !j.in[3] || !j.mult || !j.out[0] || !j.out[1] || !j.workspace) return fail(P4V_ERR_INVALID, "group: linear member with a null pointer");
            if (d->has_bias && !j.in[1]) return fail(P4V_ERR_INVALID, "group: linear member has_bias set but bias is NULL");
            return linear_impl(d, j.in[0], d->has_bias ? j.in[1] : nullptr, j.in[2], j.in[3], j.in[4], j.mult, j.out[0], j.out[1], nullptr, nullptr, c);
        }
        case P4V_JOB_MATMUL: {
            const p4v_matmul_desc* d = (const p4v_matmul_desc*)j.desc;
            if (!j.in[0] || !j.in[1] || !j.in[2] || !j.mult || !j.out[0] || !j.out[1] || !j.workspace) return fail(P4V_ERR_INVALID, "group: matmul member with a null pointer");
            return matmul_impl(d, j.in[0], j.in[1], j.in[2], j.in[3], j.mult, j.out[0], j.out[1], j.out[2], nullptr, nullptr, c);
        }
        case P4V_JOB_CONV: {
            const p4v_conv_desc* d = (const p4v_conv_desc*)j.desc;
            if (!j.in[0] || !j.in[2] || !j.in[3] || !j.mult || !j.out[0] || !j.out[1] || !j.workspace) return fail(P4V_ERR_INVALID, "group: conv member with a null pointer");
            if (d->has_bias && !j.in[1]) return fail(P4V_ERR_INVALID, "group: conv member has_bias set but bias is NULL");
            return conv_impl(d, j.in[0], d->has_bias ? j.in[1] : nullptr, j.in[2], j.in[3], j.in[4], j.mult, j.out[0], j.out[1], nullptr, nullptr, c);
        }
        default: return fail(P4V_ERR_INVALID, "group: unknown member kind %d", j.kind);
    }
}
size_t group_desc_bytes(int kind) {
    return kind == P4V_JOB_LINEAR ? sizeof(p4v_linear_desc) : kind == P4V_JOB_MATMUL ? sizeof(p4v_matmul_desc) : kind == P4V_JOB_CONV ? sizeof(p4v_conv_desc) : 0;
}

}  // namespace

// =================================================================================================
extern "C" {

int p4v_version(void) { return P4V_VERSION; }
const char* p4v_last_error(void) { return g_err.c_str(); }

size_t p4v_linear_workspace_bytes(const p4v_linear_desc* desc) {
    if (!desc) return 0;
    Ctx c{nullptr, Arena(nullptr, 0), true};   // dry run of the planner: counts, launches nothing
    float* fwd = (desc->reserved & 4) ? (float*)16 : nullptr;   // bit 2: size the workspace for quant_forward only
    if (linear_impl(desc, nullptr, nullptr, nullptr, nullptr, (const float*)1, nullptr, nullptr, nullptr, nullptr, nullptr, c, fwd) != 0) return 0;
    return c.ws.peak + 4096;
}

int p4v_linear_calibrate(const p4v_linear_desc* desc, const float* d_weight, const float* d_bias, const float* d_x,
                         const float* d_out, const float* d_grad, const float* d_mult, float* d_w_interval,
                         float* d_a_interval, float* d_scores, int32_t* d_best, void* d_workspace, size_t workspace_bytes,
                         void* stream) {
    if (!desc || !d_weight || !d_x || !d_out || !d_mult || !d_w_interval || !d_a_interval || !d_workspace)
        return fail(P4V_ERR_INVALID, "linear: null pointer");
    if (desc->has_bias && !d_bias) return fail(P4V_ERR_INVALID, "linear: has_bias set but d_bias is NULL");
    Ctx c{(hipStream_t)stream, Arena(d_workspace, workspace_bytes), false};
    return linear_impl(desc, d_weight, desc->has_bias ? d_bias : nullptr, d_x, d_out, d_grad, d_mult, d_w_interval,
                       d_a_interval, d_scores, d_best, c);
}

int p4v_linear_quant_forward(const p4v_linear_desc* desc, const float* d_weight, const float* d_bias, const float* d_x,
                             const float* d_w_interval, const float* d_a_interval, float* d_out, void* d_workspace,
                             size_t workspace_bytes, void* stream) {
    if (!desc || !d_weight || !d_x || !d_w_interval || !d_a_interval || !d_out || !d_workspace)
        return fail(P4V_ERR_INVALID, "p4v_linear_quant_forward: null pointer");
    if (desc->has_bias && !d_bias) return fail(P4V_ERR_INVALID, "p4v_linear_quant_forward: has_bias set but d_bias is null");
    Ctx c{(hipStream_t)stream, Arena(d_workspace, workspace_bytes), false};
    return linear_impl(desc, d_weight, desc->has_bias ? d_bias : nullptr, d_x, nullptr, nullptr, nullptr,
                       const_cast<float*>(d_w_interval), const_cast<float*>(d_a_interval), nullptr, nullptr, c, d_out);
}

int p4v_matmul_quant_forward(const p4v_matmul_desc* desc, const float* d_A, const float* d_B, const float* d_A_interval,
                             const float* d_B_interval, const float* d_split, float* d_out, void* d_workspace,
                             size_t workspace_bytes, void* stream) {
    if (!desc || !d_A || !d_B || !d_A_interval || !d_B_interval || !d_out || !d_workspace)
        return fail(P4V_ERR_INVALID, "p4v_matmul_quant_forward: null pointer");
    if (desc->sos && !d_split) return fail(P4V_ERR_INVALID, "p4v_matmul_quant_forward: sos needs d_split");
    Ctx c{(hipStream_t)stream, Arena(d_workspace, workspace_bytes), false};
    return matmul_impl(desc, d_A, d_B, nullptr, nullptr, nullptr, const_cast<float*>(d_A_interval),
                       const_cast<float*>(d_B_interval), const_cast<float*>(d_split), nullptr, nullptr, c, d_out);
}

size_t p4v_matmul_workspace_bytes(const p4v_matmul_desc* desc) {
    if (!desc) return 0;
    Ctx c{nullptr, Arena(nullptr, 0), true};
    float dummy = 0;
    float* fwd = (desc->reserved & 4) ? (float*)16 : nullptr;
    if (matmul_impl(desc, nullptr, nullptr, nullptr, (const float*)1, nullptr, nullptr, nullptr, &dummy, nullptr, nullptr, c, fwd) != 0) return 0;
    return c.ws.peak + 4096;
}

int p4v_matmul_calibrate(const p4v_matmul_desc* desc, const float* d_A, const float* d_B, const float* d_out,
                         const float* d_grad, const float* d_mult, float* d_A_interval, float* d_B_interval, float* d_split,
                         float* d_scores, int32_t* d_best, void* d_workspace, size_t workspace_bytes, void* stream) {
    if (!desc || !d_A || !d_B || !d_out || !d_mult || !d_A_interval || !d_B_interval || !d_workspace)
        return fail(P4V_ERR_INVALID, "matmul: null pointer");
    Ctx c{(hipStream_t)stream, Arena(d_workspace, workspace_bytes), false};
    return matmul_impl(desc, d_A, d_B, d_out, d_grad, d_mult, d_A_interval, d_B_interval, d_split, d_scores, d_best, c);
}

size_t p4v_conv_workspace_bytes(const p4v_conv_desc* desc) {
    if (!desc) return 0;
    Ctx c{nullptr, Arena(nullptr, 0), true};
    if (conv_impl(desc, nullptr, nullptr, nullptr, nullptr, (const float*)1, nullptr, nullptr, nullptr, nullptr, nullptr, c) != 0) return 0;
    return c.ws.peak + 4096;
}

int p4v_conv_calibrate(const p4v_conv_desc* desc, const float* d_weight, const float* d_bias, const float* d_x,
                       const float* d_out, const float* d_grad, const float* d_mult, float* d_w_interval, float* d_a_interval,
                       float* d_scores, int32_t* d_best, void* d_workspace, size_t workspace_bytes, void* stream) {
    if (!desc || !d_weight || !d_x || !d_out || !d_mult || !d_w_interval || !d_a_interval || !d_workspace)
        return fail(P4V_ERR_INVALID, "conv: null pointer");
    if (desc->has_bias && !d_bias) return fail(P4V_ERR_INVALID, "conv: has_bias set but d_bias is NULL");
    Ctx c{(hipStream_t)stream, Arena(d_workspace, workspace_bytes), false};
    return conv_impl(desc, d_weight, desc->has_bias ? d_bias : nullptr, d_x, d_out, d_grad, d_mult, d_w_interval, d_a_interval,
                     d_scores, d_best, c);
}


int p4v_calibrate_group(p4v_group_job* jobs, int32_t n_jobs, void* stream, void* inputs_ready_event) {
    if (n_jobs < 0 || (n_jobs > 0 && !jobs)) return fail(P4V_ERR_INVALID, "p4v_calibrate_group: bad job list");
    if (n_jobs == 0) return 0;
    for (int i = 0; i < n_jobs; ++i) {
        jobs[i].status = 0;
        if (!jobs[i].desc || group_desc_bytes(jobs[i].kind) == 0) return fail(P4V_ERR_INVALID, "p4v_calibrate_group: member %d has no descriptor / an unknown kind", i);
    }
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    cvt_probe((hipStream_t)stream);
    Group g;
    g.st = (hipStream_t)stream; g.n = n_jobs;
    g.q.resize(n_jobs); g.state.assign(n_jobs, 0); g.mirrors.resize(n_jobs);
    for (int i = 0; i < n_jobs; ++i) g.mirrors[i] = tune(TUNE_B1_PATH) == 8 ? nullptr : group_mirror(g.st, dev, i);
    g.stat_on = g_stat_on;
    g.stat_recs = g_stat_on ? &g_stat_recs : nullptr;
    g.inputs_ready = (hipEvent_t)inputs_ready_event;
    g_launch_cnt[3].fetch_add(1, std::memory_order_relaxed);
    std::vector<std::string> errs(n_jobs);
    std::atomic<long> memo_hits{0}, memo_misses{0};
    std::mutex dmu;
    std::condition_variable dcv;
    int left = n_jobs;
    for (int i = 0; i < n_jobs; ++i) {
        // members with the same descriptor run the same launches in lock step: they share the chip
        int par = 0;
        for (int k = 0; k < n_jobs; ++k)
            par += jobs[k].kind == jobs[i].kind && std::memcmp(jobs[k].desc, jobs[i].desc, group_desc_bytes(jobs[i].kind)) == 0;
        worker_pool().run([&, i, par] {
            (void)hipSetDevice(dev);
            g_stat_on = g.stat_on; g_stage = 0; g_exec_frac = 1.0; g_memo_hits = 0; g_memo_misses = 0; g_err.clear();
            Ctx c{g.st, Arena(jobs[i].workspace, jobs[i].workspace_bytes), false, &g, i, par};
            const int r = run_group_member(jobs[i], c);
            jobs[i].status = r;
            if (r) errs[i] = g_err;
            memo_hits += g_memo_hits; memo_misses += g_memo_misses;
            g_stat_on = false;
            g.finish(i);
            std::lock_guard<std::mutex> lk(dmu);
            if (--left == 0) dcv.notify_all();
        });
    }
    {
        std::unique_lock<std::mutex> lk(dmu);
        dcv.wait(lk, [&] { return left == 0; });
    }
    g_memo_hits += memo_hits.load(); g_memo_misses += memo_misses.load();
    if (tune(TUNE_PRINT) > 0) fprintf(stderr, "[p4v] group of %d: %ld launches asked for, %ld issued in %ld rounds\n", n_jobs, g.n_ops, g.n_launches, g.n_rounds);
    if (g.status) { g_err = g.err; return g.status; }
    for (int i = 0; i < n_jobs; ++i)
        if (jobs[i].status) { g_err = errs[i]; return jobs[i].status; }
    return 0;
}

int p4v_launch_counters(int64_t* out4, int reset) {
    if (out4) for (int i = 0; i < 4; ++i) out4[i] = (int64_t)g_launch_cnt[i].load(std::memory_order_relaxed);
    if (reset) for (int i = 0; i < 4; ++i) g_launch_cnt[i].store(0, std::memory_order_relaxed);
    return 0;
}

// ---- granular entry points: one part of calibration_step2 per call ----------
int p4v_amax_init_linear(const p4v_linear_desc* desc, const float* d_weight, const float* d_x, float* d_w_interval,
                         float* d_a_interval, void* d_workspace, size_t workspace_bytes, void* stream) {
    if (!desc || !d_weight || !d_x || !d_w_interval || !d_a_interval || !d_workspace)
        return fail(P4V_ERR_INVALID, "p4v_amax_init_linear: null pointer");
    Ctx c{(hipStream_t)stream, Arena(d_workspace, workspace_bytes), false};
    return linear_impl(desc, d_weight, nullptr, d_x, nullptr, nullptr, nullptr, d_w_interval, d_a_interval, nullptr, nullptr, c,
                       nullptr, Stage{ST_INIT, nullptr, nullptr});
}

int p4v_linear_search_w(const p4v_linear_desc* desc, const float* d_weight, const float* d_bias, const float* d_x,
                        const float* d_out, const float* d_grad, const float* d_w_cands, float* d_w_interval,
                        const float* d_a_interval, float* d_scores, int32_t* d_best, void* d_workspace,
                        size_t workspace_bytes, void* stream) {
    if (!desc || !d_weight || !d_x || !d_out || !d_w_cands || !d_w_interval || !d_a_interval || !d_workspace)
        return fail(P4V_ERR_INVALID, "p4v_linear_search_w: null pointer");
    if (desc->has_bias && !d_bias) return fail(P4V_ERR_INVALID, "p4v_linear_search_w: has_bias set but d_bias is NULL");
    Ctx c{(hipStream_t)stream, Arena(d_workspace, workspace_bytes), false};
    return linear_impl(desc, d_weight, desc->has_bias ? d_bias : nullptr, d_x, d_out, d_grad, nullptr, d_w_interval,
                       const_cast<float*>(d_a_interval), d_scores, d_best, c, nullptr, Stage{ST_S1, d_w_cands, nullptr});
}

int p4v_linear_search_a(const p4v_linear_desc* desc, const float* d_weight, const float* d_bias, const float* d_x,
                        const float* d_out, const float* d_grad, const float* d_a_cands, const float* d_w_interval,
                        float* d_a_interval, float* d_scores, int32_t* d_best, void* d_workspace, size_t workspace_bytes,
                        void* stream) {
    if (!desc || !d_weight || !d_x || !d_out || !d_a_cands || !d_w_interval || !d_a_interval || !d_workspace)
        return fail(P4V_ERR_INVALID, "p4v_linear_search_a: null pointer");
    if (desc->has_bias && !d_bias) return fail(P4V_ERR_INVALID, "p4v_linear_search_a: has_bias set but d_bias is NULL");
    Ctx c{(hipStream_t)stream, Arena(d_workspace, workspace_bytes), false};
    return linear_impl(desc, d_weight, desc->has_bias ? d_bias : nullptr, d_x, d_out, d_grad, nullptr,
                       const_cast<float*>(d_w_interval), d_a_interval, d_scores, d_best, c, nullptr,
                       Stage{ST_S2, nullptr, d_a_cands});
}

int p4v_amax_init_matmul(const p4v_matmul_desc* desc, const float* d_A, const float* d_B, float* d_A_interval,
                         float* d_B_interval, void* d_workspace, size_t workspace_bytes, void* stream) {
    if (!desc || !d_A || !d_B || !d_A_interval || !d_B_interval || !d_workspace)
        return fail(P4V_ERR_INVALID, "p4v_amax_init_matmul: null pointer");
    Ctx c{(hipStream_t)stream, Arena(d_workspace, workspace_bytes), false};
    float dummy_split = 0;   // only checked for presence
    return matmul_impl(desc, d_A, d_B, nullptr, nullptr, nullptr, d_A_interval, d_B_interval, &dummy_split, nullptr, nullptr, c,
                       nullptr, Stage{ST_INIT, nullptr, nullptr});
}

int p4v_matmul_search_A(const p4v_matmul_desc* desc, const float* d_A, const float* d_B, const float* d_out,
                        const float* d_grad, const float* d_A_cands, float* d_A_interval, const float* d_B_interval,
                        float* d_scores, int32_t* d_best, void* d_workspace, size_t workspace_bytes, void* stream) {
    if (!desc || !d_A || !d_B || !d_out || !d_A_cands || !d_A_interval || !d_B_interval || !d_workspace)
        return fail(P4V_ERR_INVALID, "p4v_matmul_search_A: null pointer");
    if (desc->sos) return fail(P4V_ERR_INVALID, "p4v_matmul_search_A: the split-of-softmax class searches its split (p4v_sos_search_split)");
    Ctx c{(hipStream_t)stream, Arena(d_workspace, workspace_bytes), false};
    return matmul_impl(desc, d_A, d_B, d_out, d_grad, nullptr, d_A_interval, const_cast<float*>(d_B_interval), nullptr,
                       d_scores, d_best, c, nullptr, Stage{ST_S1, d_A_cands, nullptr});
}

int p4v_sos_search_split(const p4v_matmul_desc* desc, const float* d_A, const float* d_B, const float* d_out,
                         const float* d_grad, float* d_split, float* d_A_interval, float* d_scores, int32_t* d_best,
                         void* d_workspace, size_t workspace_bytes, void* stream) {
    if (!desc || !d_A || !d_B || !d_out || !d_split || !d_A_interval || !d_workspace)
        return fail(P4V_ERR_INVALID, "p4v_sos_search_split: null pointer");
    if (!desc->sos) return fail(P4V_ERR_INVALID, "p4v_sos_search_split: descriptor is not a split-of-softmax matmul");
    Ctx c{(hipStream_t)stream, Arena(d_workspace, workspace_bytes), false};
    return matmul_impl(desc, d_A, d_B, d_out, d_grad, nullptr, d_A_interval, nullptr, d_split, d_scores, d_best, c, nullptr,
                       Stage{ST_S1, nullptr, nullptr});
}

int p4v_matmul_search_B(const p4v_matmul_desc* desc, const float* d_A, const float* d_B, const float* d_out,
                        const float* d_grad, const float* d_B_cands, const float* d_A_interval, const float* d_split,
                        float* d_B_interval, float* d_scores, int32_t* d_best, void* d_workspace, size_t workspace_bytes,
                        void* stream) {
    if (!desc || !d_A || !d_B || !d_out || !d_B_cands || !d_A_interval || !d_B_interval || !d_workspace)
        return fail(P4V_ERR_INVALID, "p4v_matmul_search_B: null pointer");
    if (desc->sos && !d_split) return fail(P4V_ERR_INVALID, "p4v_matmul_search_B: sos needs d_split");
    Ctx c{(hipStream_t)stream, Arena(d_workspace, workspace_bytes), false};
    return matmul_impl(desc, d_A, d_B, d_out, d_grad, nullptr, const_cast<float*>(d_A_interval), d_B_interval,
                       const_cast<float*>(d_split), d_scores, d_best, c, nullptr, Stage{ST_S2, nullptr, d_B_cands});
}

int p4v_amax_init_conv(const p4v_conv_desc* desc, const float* d_weight, const float* d_x, float* d_w_interval,
                       float* d_a_interval, void* d_workspace, size_t workspace_bytes, void* stream) {
    if (!desc || !d_weight || !d_x || !d_w_interval || !d_a_interval || !d_workspace)
        return fail(P4V_ERR_INVALID, "p4v_amax_init_conv: null pointer");
    Ctx c{(hipStream_t)stream, Arena(d_workspace, workspace_bytes), false};
    return conv_impl(desc, d_weight, nullptr, d_x, nullptr, nullptr, nullptr, d_w_interval, d_a_interval, nullptr, nullptr, c,
                     Stage{ST_INIT, nullptr, nullptr});
}

static int conv_search(const char* who, int want_channelwise, int stage, const p4v_conv_desc* desc, const float* d_weight,
                       const float* d_bias, const float* d_x, const float* d_out, const float* d_grad, const float* d_cands,
                       float* d_w_interval, float* d_a_interval, float* d_scores, int32_t* d_best, void* d_workspace,
                       size_t workspace_bytes, void* stream) {
    if (!desc || !d_weight || !d_x || !d_out || !d_cands || !d_w_interval || !d_a_interval || !d_workspace)
        return fail(P4V_ERR_INVALID, "%s: null pointer", who);
    if (desc->has_bias && !d_bias) return fail(P4V_ERR_INVALID, "%s: has_bias set but d_bias is NULL", who);
    if (want_channelwise >= 0 && (desc->channelwise != 0) != (want_channelwise != 0))
        return fail(P4V_ERR_INVALID, "%s: descriptor.channelwise does not match the entry point", who);
    Ctx c{(hipStream_t)stream, Arena(d_workspace, workspace_bytes), false};
    return conv_impl(desc, d_weight, desc->has_bias ? d_bias : nullptr, d_x, d_out, d_grad, nullptr, d_w_interval, d_a_interval,
                     d_scores, d_best, c, stage == ST_S1 ? Stage{ST_S1, d_cands, nullptr} : Stage{ST_S2, nullptr, d_cands});
}

int p4v_conv_search_w_channelwise(const p4v_conv_desc* desc, const float* d_weight, const float* d_bias, const float* d_x,
                                  const float* d_out, const float* d_grad, const float* d_w_cands, float* d_w_interval,
                                  const float* d_a_interval, float* d_scores, int32_t* d_best, void* d_workspace,
                                  size_t workspace_bytes, void* stream) {
    return conv_search("p4v_conv_search_w_channelwise", 1, ST_S1, desc, d_weight, d_bias, d_x, d_out, d_grad, d_w_cands,
                       d_w_interval, const_cast<float*>(d_a_interval), d_scores, d_best, d_workspace, workspace_bytes, stream);
}

int p4v_conv_search_w_layerwise(const p4v_conv_desc* desc, const float* d_weight, const float* d_bias, const float* d_x,
                                const float* d_out, const float* d_grad, const float* d_w_cands, float* d_w_interval,
                                const float* d_a_interval, float* d_scores, int32_t* d_best, void* d_workspace,
                                size_t workspace_bytes, void* stream) {
    return conv_search("p4v_conv_search_w_layerwise", 0, ST_S1, desc, d_weight, d_bias, d_x, d_out, d_grad, d_w_cands,
                       d_w_interval, const_cast<float*>(d_a_interval), d_scores, d_best, d_workspace, workspace_bytes, stream);
}

int p4v_conv_search_a(const p4v_conv_desc* desc, const float* d_weight, const float* d_bias, const float* d_x,
                      const float* d_out, const float* d_grad, const float* d_a_cands, const float* d_w_interval,
                      float* d_a_interval, float* d_scores, int32_t* d_best, void* d_workspace, size_t workspace_bytes,
                      void* stream) {
    if (desc && desc->a_bit >= 32) return fail(P4V_ERR_INVALID, "p4v_conv_search_a: a_bit >= 32 leaves the input unquantised (conv.py:600)");
    return conv_search("p4v_conv_search_a", -1, ST_S2, desc, d_weight, d_bias, d_x, d_out, d_grad, d_a_cands,
                       const_cast<float*>(d_w_interval), d_a_interval, d_scores, d_best, d_workspace, workspace_bytes, stream);
}

int p4v_score_argmax_gather(const float* d_scores, int32_t eq_n, int32_t n_blocks, const float* d_cands, float* d_interval,
                            int32_t* d_best, void* stream) {
    if (!d_scores || !d_cands || !d_interval || eq_n <= 0 || n_blocks <= 0)
        return fail(P4V_ERR_INVALID, "p4v_score_argmax_gather: bad argument");
    Ctx c{(hipStream_t)stream, Arena(nullptr, 0), false};
    SelectParams sl{d_scores, eq_n, n_blocks, d_cands, n_blocks, 1, 0, d_interval, 1, 0, nullptr, 0.0f, nullptr, 0, d_best};
    return launch_select(c, sl);
}

int p4v_quantize_i8(const float* d_x, int64_t rows, int64_t cols, int64_t cols_padded, const float* d_scales,
                    int64_t rows_per_scale, int32_t lo, int32_t hi, int8_t* d_q, void* stream) {
    if (!d_x || !d_scales || !d_q || rows <= 0 || cols <= 0 || cols_padded % 64 || cols_padded < cols || rows_per_scale <= 0)
        return fail(P4V_ERR_INVALID, "quantize_i8: bad argument");
    Ctx c{(hipStream_t)stream, Arena(nullptr, 0), false};
    PackParams p = pack2d(d_x, rows, cols, cols);
    p.Rp = (int)rows; p.Kp = (int)cols_padded; p.dst = d_q; p.C = 1;
    p.scales = d_scales; p.sc_cs = 0; p.blk_mode = 1; p.blk_div = (int)rows_per_scale;
    p.nblk_r = cdiv(rows, rows_per_scale); p.nblk_k = 1; p.lo = lo; p.hi = hi;
    return launch_pack<int8_t>(c, p);
}

int p4v_fake_quant(const float* d_x, int64_t rows, int64_t cols, const float* d_scales, int64_t rows_per_scale,
                   int32_t lo, int32_t hi, float* d_y, void* stream) {
    if (!d_x || !d_scales || !d_y || rows <= 0 || cols <= 0 || rows_per_scale <= 0)
        return fail(P4V_ERR_INVALID, "fake_quant: bad argument");
    const long n = rows * cols;
    hipLaunchKernelGGL(k_fake_quant_rows, dim3((unsigned)std::min<long>(cdiv(n, 256), 65536)), dim3(256), 0,
                       (hipStream_t)stream, d_x, (long)rows, (long)cols, d_scales, (long)rows_per_scale, (float)lo, (float)hi, d_y);
    HIPCHK(hipGetLastError());
    return 0;
}

int p4v_multi_copy(const int64_t* d_table, int32_t n, int64_t index, int64_t max_bytes, void* stream) {
    if (!d_table || n <= 0 || index < 0 || max_bytes <= 0) return fail(P4V_ERR_INVALID, "p4v_multi_copy: bad argument");
    static_assert(sizeof(long) == sizeof(int64_t), "table entries are 64-bit");
    const unsigned gx = (unsigned)std::max<long>(1, std::min<long>(cdiv(max_bytes / 16, 256 * 8), 256));
    hipLaunchKernelGGL(k_multi_copy, dim3(gx, (unsigned)n), dim3(256), 0, (hipStream_t)stream, (const long*)d_table, (long)index);
    HIPCHK(hipGetLastError());
    return 0;
}

// What an event pair measures around NOTHING on an idle stream: the two timestamp packets and the gap between them.  The same
// cost sits inside every timed launch (6-9 us against a rocprofv3 kernel trace of the same launches on MI355X / ROCm 7.2), so
// it is measured when timing is switched on -- minimum over 32 empty pairs on the null stream -- and subtracted from every
// record.  Reported as p4v_kernel_stats.event_overhead_ms; tools/prof_join.py shows the corrected averages next to the trace's.
thread_local double g_evt_overhead_ms = 0.0;
int p4v_stats_enable(int enable) {
    g_stat_on = enable != 0;
    if (g_stat_on) {
        hipEvent_t a, b;
        HIPCHK(hipEventCreate(&a));
        HIPCHK(hipEventCreate(&b));
        double best = 1e9;
        for (int i = 0; i < 32; ++i) {
            HIPCHK(hipEventRecord(a, nullptr));
            HIPCHK(hipEventRecord(b, nullptr));
            HIPCHK(hipEventSynchronize(b));
            float ms = 0;
            HIPCHK(hipEventElapsedTime(&ms, a, b));
            best = std::min(best, (double)ms);
        }
        hipEventDestroy(a);
        hipEventDestroy(b);
        g_evt_overhead_ms = best < 1e8 ? best : 0.0;
    }
    return 0;
}

static int stats_drain() {
    for (auto& r : g_stat_recs) {
        HIPCHK(hipEventSynchronize(r.b));
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, r.a, r.b));
        ms = (float)std::max(0.0, (double)ms - g_evt_overhead_ms);
        if (r.kind == 0 || (r.kind >= 2 && r.kind <= 9) || (r.kind >= 12 && r.kind <= 14)) { g_stats.sweep_i8_ms += ms; g_stats.sweep_i8_launches++; g_stats.sweep_i8_macs += r.macs; g_stats.sweep_i8_alg_macs += r.alg; }
        if (r.kind == 2) { g_stats.sweep6_ms += ms; g_stats.sweep6_launches++; g_stats.sweep6_macs += r.macs; g_stats.sweep6_alg_macs += r.alg; }
        if (r.kind == 3 || r.kind == 4) { g_stats.sweep7_ms += ms; g_stats.sweep7_launches++; g_stats.sweep7_macs += r.macs; g_stats.sweep7_alg_macs += r.alg; }
        if (r.kind == 4) { g_stats.sweep7_twin_ms += ms; g_stats.sweep7_twin_launches++; }
        if (r.kind == 1 || r.kind == 11) { g_stats.sweep_f32_ms += ms; g_stats.sweep_f32_launches++; g_stats.sweep_f32_macs += r.macs; g_stats.sweep_f32_alg_macs += r.alg; }
        hipEventDestroy(r.a);
        hipEventDestroy(r.b);
        r.ms = ms;
        g_stat_done.push_back(r);
    }
    g_stat_recs.clear();
    return 0;
}

int p4v_stats_reset(void) {
    int r = stats_drain();
    g_stats = p4v_kernel_stats{};
    g_stat_done.clear();
    g_memo_hits = 0; g_memo_misses = 0;
    return r;
}

int p4v_stats_get(p4v_kernel_stats* out) {
    if (!out) return fail(P4V_ERR_INVALID, "stats_get: null");
    int r = stats_drain();
    g_stats.memo_hits = g_memo_hits; g_stats.memo_misses = g_memo_misses;
    g_stats.event_overhead_ms = g_evt_overhead_ms;
    *out = g_stats;
    return r;
}

int p4v_stats_launches(p4v_launch_record* out, int64_t capacity, int64_t* count) {
    int r = stats_drain();
    if (r) return r;
    if (count) *count = (int64_t)g_stat_done.size();
    if (out)
        for (int64_t i = 0; i < capacity && i < (int64_t)g_stat_done.size(); ++i) {
            const StatRec& s = g_stat_done[(size_t)i];
            out[i] = p4v_launch_record{s.kind, s.stage, s.gx, s.gz, (double)s.ms, 2.0 * s.macs, 2.0 * s.alg, s.bytes};
        }
    return 0;
}

int p4v_prune_counters(int64_t* out4, int reset) {
    if (out4) for (int i = 0; i < 4; ++i) out4[i] = (int64_t)g_prune_cnt[i].load(std::memory_order_relaxed);
    if (reset) for (int i = 0; i < 4; ++i) g_prune_cnt[i].store(0, std::memory_order_relaxed);
    return 0;
}

int p4v_debug_set_variant(int variant, int force_generic) {
    if (variant < 0 || variant >= (1 << 30)) return fail(P4V_ERR_INVALID, "p4v_debug_set_variant: bad variant word");
    g_variant_word.store(variant | (force_generic ? (1 << 30) : 0), std::memory_order_relaxed);
    return 0;
}

int p4v_debug_set_tuning(int key, int value) {
    if (key < 0 || key >= 16) return fail(P4V_ERR_INVALID, "p4v_debug_set_tuning: unknown key %d", key);
    g_tune[key].store(value, std::memory_order_relaxed);
    return 0;
}

int p4v_debug_topk_rows(const float* d_mass, int segs, int n, int k, int32_t* d_idx, void* stream) {
    if (!d_mass || !d_idx || segs <= 0 || n <= 0 || k <= 0 || k > n) return fail(P4V_ERR_INVALID, "p4v_debug_topk_rows: bad argument");
    hipLaunchKernelGGL(k_topk_rows, dim3(segs), dim3(1024), 0, (hipStream_t)stream, TopkParams{d_mass, n, k, d_idx});
    HIPCHK(hipGetLastError());
    return 0;
}

int p4v_pack_plane_i8(const p4v_plane_desc* d, const float* d_x, const float* d_scales, int8_t* d_q, void* stream) {
    if (!d || !d_x || !d_q || d->rows <= 0 || d->cols <= 0 || d->cols_padded % 64 || d->cols_padded < d->cols)
        return fail(P4V_ERR_INVALID, "p4v_pack_plane_i8: bad argument");
    if (d->mode != P4V_PLANE_SYM && d->mode != P4V_PLANE_SOS_HI && d->mode != P4V_PLANE_SOS_LO && d->mode != P4V_PLANE_TWIN)
        return fail(P4V_ERR_INVALID, "p4v_pack_plane_i8: unknown mode %d", d->mode);
    if (d->mode == P4V_PLANE_TWIN && (d->lo > 0 || d->hi < 0 || !(d->const_scale > 0.0f)))
        return fail(P4V_ERR_INVALID, "p4v_pack_plane_i8: the merged twin plane needs lo <= 0 <= hi and const_scale > 0");
    if (d->mode != P4V_PLANE_SYM && !d_scales) return fail(P4V_ERR_INVALID, "p4v_pack_plane_i8: split-of-softmax planes need d_scales = &split");
    if (d->mode == P4V_PLANE_SYM && d_scales && d->rows_per_scale <= 0) return fail(P4V_ERR_INVALID, "p4v_pack_plane_i8: rows_per_scale");
    if (d->lo < -128 || d->hi > 127 || d->qmax < 2 || d->qmax > 128) return fail(P4V_ERR_INVALID, "p4v_pack_plane_i8: grid wider than int8");
    Ctx c{(hipStream_t)stream, Arena(nullptr, 0), false};
    cvt_bias(c);     // (first use of the process: the conversion quant16_sat8 relies on is probed, into a scratch of the library's own)
    PackParams p = pack2d(d_x, d->rows, d->cols, d->cols);
    p.Rp = (int)d->rows; p.Kp = (int)d->cols_padded; p.dst = d_q; p.C = 1;
    p.scales = d_scales; p.sc_cs = 0; p.neg_scale = d->const_scale;
    p.lo = d->lo; p.hi = d->hi; p.qm1 = (float)(d->qmax - 1);
    if (d->mode == P4V_PLANE_SYM) {
        p.mode = PACK_SYM;
        if (d_scales) { p.blk_mode = 1; p.blk_div = (int)d->rows_per_scale; p.nblk_r = cdiv(d->rows, d->rows_per_scale); p.nblk_k = 1; }
    } else if (d->mode == P4V_PLANE_TWIN) {
        p.mode = PACK_TWIN_I8;
    } else {
        p.mode = d->mode == P4V_PLANE_SOS_HI ? PACK_SOS_HI : PACK_SOS_LO;
    }
    return launch_pack<int8_t>(c, p);
}

int p4v_export_quantize(const p4v_export_desc* d, const float* d_src, const float* d_scale1, const float* d_scale2,
                        void* d_dst, void* stream) {
    if (!d || !d_src || !d_scale1 || !d_dst) return fail(P4V_ERR_INVALID, "p4v_export_quantize: null pointer");
    if (d->mode < P4V_EXPORT_SYM_I8 || d->mode > P4V_EXPORT_SOS_U8) return fail(P4V_ERR_INVALID, "p4v_export_quantize: unknown mode %d", d->mode);
    if (d->mode == P4V_EXPORT_SOS_U8 && !d_scale2) return fail(P4V_ERR_INVALID, "p4v_export_quantize: the split-of-softmax format needs d_scale2 = A_interval");
    ExportParams p{};
    long total = 1;
    for (int i = 0; i < 4; ++i) {
        if (d->dims[i] <= 0 || d->scale1_div[i] <= 0 || (d_scale2 && d->scale2_div[i] <= 0))
            return fail(P4V_ERR_INVALID, "p4v_export_quantize: non-positive dimension / block size");
        p.d[i] = d->dims[i]; p.ss[i] = d->src_stride[i];
        p.s1s[i] = d->scale1_stride[i]; p.s1d[i] = d->scale1_div[i];
        p.s2s[i] = d->scale2_stride[i]; p.s2d[i] = d_scale2 ? d->scale2_div[i] : 1;
        total *= d->dims[i];
    }
    p.src = d_src; p.s1 = d_scale1; p.s2 = d_scale2; p.s2_const = d->scale2_const;
    p.mode = d->mode == P4V_EXPORT_SYM_I8 ? EXP_SYM_I8 : d->mode == P4V_EXPORT_SYM_F32 ? EXP_SYM_F32
           : d->mode == P4V_EXPORT_GELU_U8 ? EXP_GELU_U8 : EXP_SOS_U8;
    p.lo1 = d->lo1; p.hi1 = d->hi1; p.lo2 = d->lo2; p.hi2 = d->hi2; p.qm1 = (float)(d->qmax - 1);
    p.dst = d_dst;
    const unsigned blocks = (unsigned)std::min<long>(cdiv(total, 256), 256L * 32);
    hipLaunchKernelGGL(k_export, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
    HIPCHK(hipGetLastError());
    return 0;
}

}  // extern "C"
