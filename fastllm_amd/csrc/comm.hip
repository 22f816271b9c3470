// comm.hip -- host side of the tensor-parallel group's small collectives: the peer-mapped inbox and LL regions
// (allocation, hipIpc export / connect, peer pointers), the bootstrap over an RCCL communicator (handle all-gather,
// self-tests against known sums, all-ranks vote) and the one-shot launch helper.  Device side: k_comm.hip, comm_ll.h.
#include "model.h"

#include <math.h>
#include <stdarg.h>
#include <stdlib.h>

#include <algorithm>
#include <array>
#include <memory>
#include <mutex>
#include <utility>
#include <vector>

namespace fl {

// ------------------------------------------------------------------------------- one-shot collectives
constexpr size_t kCommFlagBytes = 16384;            // one-workgroup collectives: words [0, tp); many-workgroup ones: [64 + 8 g + rank], g < 256 (k_comm.hip)
static int comm_connect_impl(Model *m, const void *handles);
static int comm_export_impl(Shard &sh, void *handle_out);

// Handles exported by models of THIS process.  A process may host several ranks of one group (tests and rehearsals of an 8-rank
// group on a one-GPU box, which admits fewer GPU processes than that: 4 processes x 2 ranks); hipIpcOpenMemHandle does not open
// a handle in the process that exported it, so such a peer is reached through its plain device pointer instead.
static std::mutex g_exported_mu;
static std::vector<std::pair<std::array<char, FL_IPC_HANDLE_BYTES>, void *>> g_exported;
static void *comm_exported_here(const void *handle) {
    std::lock_guard<std::mutex> lock(g_exported_mu);
    for (auto &e : g_exported) if (!memcmp(e.first.data(), handle, FL_IPC_HANDLE_BYTES)) return e.second;
    return nullptr;
}
void comm_forget(void *local) {
    std::lock_guard<std::mutex> lock(g_exported_mu);
    g_exported.erase(std::remove_if(g_exported.begin(), g_exported.end(), [&](auto &e) { return e.second == local; }), g_exported.end());
}

// hipDeviceMallocUncached buffers are NEVER handed back to the runtime.  Found in round 5 (tools/seq_probe.py): after a model with an
// uncached inbox had been destroyed -- hipFree of the inbox -- later models of the same process, single-GPU ones included, computed
// garbage (10 of 12 models wrong, deterministically; with the inbox from plain hipMalloc, or never freed, 0 of 12): on this stack
// (ROCm 7.2, dmabuf IPC) memory that was once mapped uncached and freed comes back to later allocations in a state that corrupts them.
// A destroyed shard's inbox (1-3 MB) goes to a free list instead and serves the next shard that asks for that device and size.
struct InboxBuf { int device; size_t bytes; void *p; };
static std::mutex g_inbox_mu;
static std::vector<InboxBuf> g_inbox_free;
static int inbox_acquire(int device, size_t bytes, void **out) {
    {
        std::lock_guard<std::mutex> lock(g_inbox_mu);
        for (size_t i = 0; i < g_inbox_free.size(); i++)
            if (g_inbox_free[i].device == device && g_inbox_free[i].bytes == bytes) {
                *out = g_inbox_free[i].p;
                g_inbox_free.erase(g_inbox_free.begin() + (long)i);
                return FL_OK;
            }
    }
    FL_HIP(hipExtMallocWithFlags(out, bytes, hipDeviceMallocUncached));
    return FL_OK;
}
void comm_inbox_release(int device, size_t bytes, void *p) {
    if (!p) return;
    std::lock_guard<std::mutex> lock(g_inbox_mu);
    g_inbox_free.push_back(InboxBuf{device, bytes, p});
}

int comm_alloc(Model *m, Shard &sh) {
    PeerComm &pc = sh.pc;
    FL_HIP(hipSetDevice(sh.device));
    // a slot holds at least one rank's share of the logits, so the per-step gather is always a one-shot collective
    pc.nmax = (std::max<int64_t>(std::max<int64_t>(tune(TK_AR_INBOX_FLOATS), sh.Vs), 4) + 3) / 4 * 4;
    pc.ll_off = (kCommFlagBytes + (size_t)2 * m->tp * pc.nmax * 4 + 255) & ~(size_t)255;
    pc.bytes = pc.ll_off + (size_t)2 * m->tp * m->D.h * 8;
    FL_TRY(inbox_acquire(sh.device, pc.bytes, &pc.local));
    FL_HIP(hipMemset(pc.local, 0, pc.bytes));
    FL_HIP(hipMalloc((void **)&pc.epoch, 64));
    FL_HIP(hipMemset(pc.epoch, 0, 64));
    FL_HIP(hipMalloc((void **)&pc.ll_dev, sizeof(LLTable)));
    FL_HIP(hipHostMalloc((void **)&pc.err, 64, hipHostMallocDefault));
    *pc.err = 0;
    pc.timeout_ticks = (long long)tune(TK_AR_TIMEOUT_MS) * 100000LL;       // 100 MHz wall clock
    FL_HIP(hipDeviceSynchronize());
    m->hbm_bytes += (int64_t)pc.bytes;
    return FL_OK;
}

void comm_set_entry(PeerComm &pc, int r, void *base) {
    pc.tab.flags[r] = (uint32_t *)base;
    pc.tab.inbox[r] = (float *)((char *)base + kCommFlagBytes);
    pc.ll.peer[r] = (uint64_t *)((char *)base + pc.ll_off);
}

// every entry of the table is set: hand the fused all-reduce's view of the group to the device
int comm_ll_publish(Model *m, Shard &sh) {
    PeerComm &pc = sh.pc;
    FL_HIP(hipSetDevice(sh.device));
    pc.ll.epoch_ctr = pc.epoch; pc.ll.err = pc.err; pc.ll.abort_flag = pc.epoch + 8; pc.ll.timeout_ticks = pc.timeout_ticks;
    pc.ll.rank = sh.rank; pc.ll.tp = m->tp; pc.ll.n = (int)m->D.h; pc.ll.slots = (int)(2 * m->D.L); pc.ll.loop = pc.loopback ? 1 : 0;
    FL_HIP(hipMemcpy(pc.ll_dev, &pc.ll, sizeof(LLTable), hipMemcpyHostToDevice));
    pc.ll_ok = true;
    return FL_OK;
}

int comm_ipc_export(Model *m, void *handle_out) {
    if (!m || !handle_out) FL_FAIL(FL_ERR_BAD_ARGUMENT, "null argument");
    if (m->tp_mode != FL_TP_MULTI_PROCESS || m->tp < 2) FL_FAIL(FL_ERR_BAD_ARGUMENT, "peer inboxes exist in FL_TP_MULTI_PROCESS mode only");
    return comm_export_impl(m->shards[0], handle_out);
}

static int comm_export_impl(Shard &sh, void *handle_out) {
    FL_HIP(hipSetDevice(sh.device));
    static_assert(sizeof(hipIpcMemHandle_t) <= FL_IPC_HANDLE_BYTES, "handle size");
    hipIpcMemHandle_t h;
    FL_HIP(hipIpcGetMemHandle(&h, sh.pc.local));
    memset(handle_out, 0, FL_IPC_HANDLE_BYTES);
    memcpy(handle_out, &h, sizeof h);
    comm_forget(sh.pc.local);
    std::array<char, FL_IPC_HANDLE_BYTES> key;
    memcpy(key.data(), handle_out, FL_IPC_HANDLE_BYTES);
    std::lock_guard<std::mutex> lock(g_exported_mu);
    g_exported.emplace_back(key, sh.pc.local);
    return FL_OK;
}

int comm_ipc_connect(Model *m, const void *handles) {
    if (!m || !handles) FL_FAIL(FL_ERR_BAD_ARGUMENT, "null argument");
    if (m->tp_mode != FL_TP_MULTI_PROCESS || m->tp < 2) FL_FAIL(FL_ERR_BAD_ARGUMENT, "peer inboxes exist in FL_TP_MULTI_PROCESS mode only");
    std::lock_guard<std::mutex> lock(m->mu);
    return comm_connect_impl(m, handles);
}

static int comm_connect_impl(Model *m, const void *handles) {
    Shard &sh = m->shards[0];
    PeerComm &pc = sh.pc;
    if (pc.connected) FL_FAIL(FL_ERR_BAD_ARGUMENT, "peer inboxes are already connected");
    FL_HIP(hipSetDevice(sh.device));
    for (int r = 0; r < m->tp; r++) {
        if (r == sh.rank) { comm_set_entry(pc, r, pc.local); continue; }
        if (void *here = comm_exported_here((const char *)handles + (size_t)r * FL_IPC_HANDLE_BYTES)) {
            // a rank hosted by this same process (on this same GPU or a peer-accessible one): no mapping to open or close
            hipPointerAttribute_t at;
            if (hipPointerGetAttributes(&at, here) == hipSuccess) {
                if (at.device == sh.device) pc.shares_device = true;
                else {
                    const hipError_t pe = hipDeviceEnablePeerAccess(at.device, 0);
                    if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) { (void)hipGetLastError(); FL_FAIL(FL_ERR_HIP, "no peer access to rank %d's GPU: %s", r, hipGetErrorString(pe)); }
                    (void)hipGetLastError();
                }
            } else (void)hipGetLastError();
            comm_set_entry(pc, r, here);
            continue;
        }
        hipIpcMemHandle_t h;
        memcpy(&h, (const char *)handles + (size_t)r * FL_IPC_HANDLE_BYTES, sizeof h);
        void *p = nullptr;
        hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            for (void *&mp : pc.mapped) if (mp) { (void)hipIpcCloseMemHandle(mp); mp = nullptr; }
            FL_FAIL(FL_ERR_HIP, "hipIpcOpenMemHandle of rank %d's inbox failed: %s", r, hipGetErrorString(e));
        }
        pc.mapped[r] = p;
        comm_set_entry(pc, r, p);
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, p) == hipSuccess) pc.shares_device = pc.shares_device || at.device == sh.device;
        else (void)hipGetLastError();
    }
    pc.connected = true;
    return comm_ll_publish(m, sh);
}

// TK_DEBUG_TP_LOOPBACK: one rank of a tp-way group alone on its GPU, every inbox entry its own: the kernels push what the tp ranks
// would push (CommTable::loop, LLTable::loop) and find it at once, so the rank's step runs with its exchange code in place and
// no peer to wait for.  What it measures: the rank's own time per step.  What it computes: nothing meaningful (sums of tp copies).
int comm_connect_loopback(Model *m) {
    Shard &sh = m->shards[0];
    PeerComm &pc = sh.pc;
    for (int r = 0; r < m->tp; r++) comm_set_entry(pc, r, pc.local);
    pc.tab.loop = 1;
    pc.loopback = true;
    pc.connected = true;
    return comm_ll_publish(m, sh);
}

// n floats in chunks of at most nmax; reduce: out = sum over ranks (in == out allowed);
// gather: out[r * out_stride + i] = in_r[i]
int oneshot(Model *m, Shard &sh, bool gather, const float *in, float *out, int64_t n, int64_t out_stride, hipStream_t on) {
    PeerComm &pc = sh.pc;
    Launcher L = make_launcher(m, sh);
    if (on) L.stream = on;
    // workgroups of a large message (FL_ONESHOT_WIDE: 0 one, 1 up to 256, N > 1 up to N).  They WAIT for the same slice of every peer, so all
    // of them, of every rank, must be resident at once.  One rank per GPU: always so.  Ranks that share a card (a test rig): workgroups of
    // different processes do not share a CU on this hardware -- the fourth process' 80 workgroups found no CU while three times 80 of the others
    // sat waiting for them (4 x 64 = 256 CUs ran, measured) -- so the ranks of a card split 192 CUs between them and leave the rest to whichever rank is still computing
    const int64_t wide = tune(TK_ONESHOT_WIDE);
    const int max_wgs = wide <= 0 || !pc.wide_ok ? 0 : pc.shares_device ? std::max(1, 192 / m->tp) : wide > 1 ? (int)std::min<int64_t>(wide, 256) : 256;
    for (int64_t off = 0; off < n; off += pc.nmax) {
        const int64_t c = std::min(pc.nmax, n - off);
        FL_TRY(launch_oneshot(L, gather, in + off, out + off, pc.tab, sh.rank, m->tp, c, pc.nmax, out_stride, pc.epoch, pc.err, pc.timeout_ticks, pc.epoch + 8,
                              pc.epoch + 12, max_wgs));
    }
    return FL_OK;
}

// One all-reduce of n integer-valued floats over the connected inboxes -- x_r[i] = (i % 251) + 1000 r: exact in fp32 whatever the
// summation order -- checked against the closed form, with a 2 s bound on every wait (a path that does not work must fail fast).
// n picks the form (k_comm.hip: 16 384 floats and more run on many workgroups).  Collective: every rank of the group calls it.
// *good = 0: this rank saw a wrong sum or a wait that gave up; the error word and the abort flag are cleared again either way (a wait
// that gave up still moved the epoch on every rank), so the group stays usable on the forms that did pass.
int comm_prove_oneshot(Model *m, Shard &sh, int64_t n, int *good) {
    PeerComm &pc = sh.pc;
    *good = 0;
    if (!pc.connected || n < 1 || n > pc.nmax) return FL_OK;
    const int tp = m->tp;
    FL_HIP(hipSetDevice(sh.device));
    float *w = nullptr;
    if (hipMalloc((void **)&w, (size_t)n * 4) != hipSuccess) { (void)hipGetLastError(); return FL_OK; }
    std::vector<float> x((size_t)n);
    for (int64_t i = 0; i < n; i++) x[(size_t)i] = (float)(i % 251) + 1000.f * sh.rank;
    const long long keep = pc.timeout_ticks;
    int ok = hipMemcpyAsync(w, x.data(), (size_t)n * 4, hipMemcpyHostToDevice, sh.stream) == hipSuccess;
    pc.timeout_ticks = 200000000LL;
    ok = ok && oneshot(m, sh, false, w, w, n, 0) == FL_OK;
    pc.timeout_ticks = keep;
    ok = ok && hipMemcpyAsync(x.data(), w, (size_t)n * 4, hipMemcpyDeviceToHost, sh.stream) == hipSuccess;
    ok = hipStreamSynchronize(sh.stream) == hipSuccess && ok;
    ok = ok && *pc.err == 0;
    const float base = 1000.f * (float)(tp * (tp - 1) / 2);
    for (int64_t i = 0; i < n && ok; i++) ok = x[(size_t)i] == (float)tp * (float)(i % 251) + base;
    (void)hipFree(w);
    *pc.err = 0;
    (void)hipMemsetAsync(pc.epoch + 8, 0, 4, sh.stream);          // the abort word a failed wait left behind
    (void)hipMemsetAsync(pc.epoch + 12, 0, 4, sh.stream);         // ... and the many-workgroup form's ticket
    (void)hipStreamSynchronize(sh.stream);
    (void)hipGetLastError();
    *good = ok ? 1 : 0;
    return FL_OK;
}

// fl_comm_selftest: the proof above on a connected group, whoever connected it (RCCL bootstrap or fl_comm_ipc_connect)
int comm_selftest(Model *m, int64_t n, int *ok) {
    if (!m || !ok) FL_FAIL(FL_ERR_BAD_ARGUMENT, "comm_selftest: null argument");
    *ok = 0;
    if (m->tp_mode != FL_TP_MULTI_PROCESS || m->shards.size() != 1) FL_FAIL(FL_ERR_UNSUPPORTED, "comm_selftest: one rank of an FL_TP_MULTI_PROCESS group");
    std::lock_guard<std::mutex> lock(m->mu);
    Shard &sh = m->shards[0];
    if (!sh.pc.connected) FL_FAIL(FL_ERR_UNSUPPORTED, "comm_selftest: the peer inboxes are not connected");
    if (n < 1 || n > sh.pc.nmax) FL_FAIL(FL_ERR_BAD_ARGUMENT, "comm_selftest: 1 ... %lld values (FL_AR_INBOX_FLOATS)", (long long)sh.pc.nmax);
    return comm_prove_oneshot(m, sh, n, ok);
}

// Self-contained bootstrap when an RCCL communicator exists: all-gather the IPC handles through it,
// map the peers, then prove the path on integer-valued data against ncclAllReduce.  Every decision
// is agreed by all ranks (min over ranks), so either all use the one-shot path or none does.
int comm_bootstrap_over_rccl(Model *m) {
    Shard &sh = m->shards[0];
    PeerComm &pc = sh.pc;
    const int tp = m->tp;
    const bool verbose = tune(TK_VERBOSE) != 0;
    FL_HIP(hipSetDevice(sh.device));
    char *dbuf = nullptr;
    const size_t test_n = 4096;
    FL_HIP(hipMalloc((void **)&dbuf, (size_t)tp * FL_IPC_HANDLE_BYTES + 64 + 2 * test_n * 4));
    std::vector<char> hbuf((size_t)tp * FL_IPC_HANDLE_BYTES);
    int ok = comm_export_impl(sh, hbuf.data() + (size_t)sh.rank * FL_IPC_HANDLE_BYTES) == FL_OK;
    auto agree = [&](int mine, int *all) -> int {
        int *d = (int *)(dbuf + (size_t)tp * FL_IPC_HANDLE_BYTES);
        FL_HIP(hipMemcpyAsync(d, &mine, 4, hipMemcpyHostToDevice, sh.stream));
        FL_NCCL(ncclAllReduce(d, d, 1, ncclInt, ncclMin, sh.comm, sh.stream));
        FL_HIP(hipMemcpyAsync(all, d, 4, hipMemcpyDeviceToHost, sh.stream));
        FL_HIP(hipStreamSynchronize(sh.stream));
        return FL_OK;
    };
    auto finish = [&](int rc) { (void)hipFree(dbuf); return rc; };
    int all = 0;
    if (agree(ok, &all) != FL_OK) return finish(FL_ERR_RCCL);
    if (!all) return finish(FL_OK);
    FL_HIP(hipMemcpyAsync(dbuf + (size_t)sh.rank * FL_IPC_HANDLE_BYTES, hbuf.data() + (size_t)sh.rank * FL_IPC_HANDLE_BYTES,
                          FL_IPC_HANDLE_BYTES, hipMemcpyHostToDevice, sh.stream));
    FL_NCCL(ncclAllGather(dbuf + (size_t)sh.rank * FL_IPC_HANDLE_BYTES, dbuf, FL_IPC_HANDLE_BYTES, ncclChar, sh.comm, sh.stream));
    FL_HIP(hipMemcpyAsync(hbuf.data(), dbuf, hbuf.size(), hipMemcpyDeviceToHost, sh.stream));
    FL_HIP(hipStreamSynchronize(sh.stream));
    ok = comm_connect_impl(m, hbuf.data()) == FL_OK;
    if (!ok && verbose) fprintf(stderr, "[fastllm_mi355x] rank %d: %s\n", sh.rank, last_error());
    if (agree(ok, &all) != FL_OK) return finish(FL_ERR_RCCL);
    if (all) {
        // proof run: x_r[i] = (i % 251) + 1000 r  (exact in fp32 whatever the summation order)
        float *a = (float *)(dbuf + (size_t)tp * FL_IPC_HANDLE_BYTES + 64), *b = a + test_n;
        std::vector<float> x(test_n), ya(test_n), yb(test_n);
        for (size_t i = 0; i < test_n; i++) x[i] = (float)(i % 251) + 1000.f * sh.rank;
        FL_HIP(hipMemcpyAsync(a, x.data(), test_n * 4, hipMemcpyHostToDevice, sh.stream));
        FL_HIP(hipMemcpyAsync(b, x.data(), test_n * 4, hipMemcpyHostToDevice, sh.stream));
        const long long keep = pc.timeout_ticks;
        pc.timeout_ticks = 200000000LL;              // 2 s: a path that does not work must fail fast here
        // the exchange fused into the GEMV epilogues first (epoch counter still 0: it uses epoch 1, never seen again)
        {
            const size_t ll_n = std::min<size_t>(test_n, (size_t)m->D.h);
            int ll_good = comm_ll_publish(m, sh) == FL_OK;
            if (ll_good) {
                Launcher L = make_launcher(m, sh);
                ll_good = launch_ll_allreduce(L, pc.ll_dev, 1, a, a, (int64_t)ll_n) == FL_OK;
                FL_HIP(hipMemcpyAsync(ya.data(), a, ll_n * 4, hipMemcpyDeviceToHost, sh.stream));
                FL_HIP(hipStreamSynchronize(sh.stream));
                ll_good = ll_good && *pc.err == 0;
                for (size_t i = 0; i < ll_n && ll_good; i++) ll_good = ya[i] == (float)tp * (float)(i % 251) + 1000.f * (float)(tp * (tp - 1) / 2);
                FL_HIP(hipMemcpyAsync(a, x.data(), test_n * 4, hipMemcpyHostToDevice, sh.stream));
            }
            int ll_all = 0;
            if (agree(ll_good, &ll_all) != FL_OK) return finish(FL_ERR_RCCL);
            if (!ll_good && verbose) fprintf(stderr, "[fastllm_mi355x] rank %d: fused all-reduce self-test failed (err 0x%x)\n", sh.rank, *pc.err);
            *pc.err = 0;
            FL_HIP(hipMemsetAsync(pc.epoch + 8, 0, 4, sh.stream));           // the abort word a failed wait left behind
            FL_HIP(hipStreamSynchronize(sh.stream));
            pc.timeout_ticks = keep;
            if (ll_all) { if (comm_ll_publish(m, sh) != FL_OK) return finish(FL_ERR_HIP); }    // with the real timeout
            pc.ll_ok = ll_all != 0;
            pc.timeout_ticks = 200000000LL;
        }
        int rc = oneshot(m, sh, false, a, a, (int64_t)test_n, 0);
        pc.timeout_ticks = keep;
        if (rc != FL_OK) return finish(rc);
        FL_NCCL(ncclAllReduce(b, b, test_n, ncclFloat, ncclSum, sh.comm, sh.stream));
        FL_HIP(hipMemcpyAsync(ya.data(), a, test_n * 4, hipMemcpyDeviceToHost, sh.stream));
        FL_HIP(hipMemcpyAsync(yb.data(), b, test_n * 4, hipMemcpyDeviceToHost, sh.stream));
        FL_HIP(hipStreamSynchronize(sh.stream));
        ok = *pc.err == 0 && memcmp(ya.data(), yb.data(), test_n * 4) == 0;
        if (!ok && verbose) fprintf(stderr, "[fastllm_mi355x] rank %d: one-shot all-reduce self-test failed (err 0x%x)\n", sh.rank, *pc.err);
        if (agree(ok, &all) != FL_OK) return finish(FL_ERR_RCCL);
        if (all && tune(TK_ONESHOT_WIDE) > 0 && pc.nmax >= 16384) {
            // the many-workgroup form of the same collective (messages of 16 384 floats and more: a batch's [B, h], short prompts) is
            // proved on its own; if it fails anywhere, every rank keeps the one-workgroup form
            int wide_good = 0, wide_all = 0;
            if (comm_prove_oneshot(m, sh, std::min<int64_t>(pc.nmax, 98304) / 4 * 4, &wide_good) != FL_OK) wide_good = 0;
            if (agree(wide_good, &wide_all) != FL_OK) return finish(FL_ERR_RCCL);
            if (!wide_good && verbose) fprintf(stderr, "[fastllm_mi355x] rank %d: many-workgroup all-reduce self-test failed\n", sh.rank);
            pc.wide_ok = wide_all != 0;
        }
    }
    if (!all) {                                      // stay on RCCL; the epoch counters may differ now, so the path is closed for good
        pc.connected = false;
        pc.ll_ok = false;
        *pc.err = 0;
        FL_HIP(hipMemsetAsync(pc.epoch + 8, 0, 4, sh.stream));
        FL_HIP(hipStreamSynchronize(sh.stream));
    }
    if (verbose && sh.rank == 0)
        fprintf(stderr, "[fastllm_mi355x] small collectives: %s%s%s\n", all ? "one-shot over peer-mapped HBM" : "RCCL", all && pc.ll_ok ? ", all-reduce fused into the GEMV epilogues" : "",
                all && !pc.wide_ok ? ", one workgroup per collective" : "");
    return finish(FL_OK);
}

// fl_comm_probe: `iters` all-reduces of n floats back to back on the shard's stream, HIP events around the batch.
int comm_probe(Model *m, int form, int64_t n, int iters, double *us_per_call) {
    if (!m || !us_per_call || iters < 1 || iters > 100000 || n < 1) FL_FAIL(FL_ERR_BAD_ARGUMENT, "comm_probe: bad arguments");
    *us_per_call = -1.0;
    if (m->tp_mode != FL_TP_MULTI_PROCESS || m->shards.size() != 1) return FL_OK;            // one process per GPU only
    std::lock_guard<std::mutex> lock(m->mu);
    Shard &sh = m->shards[0];
    PeerComm &pc = sh.pc;
    if (n > m->D.h) FL_FAIL(FL_ERR_BAD_ARGUMENT, "comm_probe: at most hidden_size values");
    if ((form == 0 && !sh.comm) || (form == 1 && (!pc.connected || n > pc.nmax)) || (form == 2 && (!pc.connected || !pc.ll_ok || pc.shares_device)) || form < 0 || form > 2)
        return FL_OK;
    FL_HIP(hipSetDevice(sh.device));
    float *buf = sh.dec.delta;                                   // [h] fp32 scratch of the decode step
    FL_HIP(hipMemsetAsync(buf, 0, (size_t)n * 4, sh.stream));
    struct Events {                                         // destroyed on every way out (an FL_HIP below returns early)
        hipEvent_t e[3] = {nullptr, nullptr, nullptr};
        ~Events() { for (auto x : e) if (x) (void)hipEventDestroy(x); }
    } ev;
    for (auto &x : ev.e) FL_HIP(hipEventCreate(&x));
    hipEvent_t e0 = ev.e[0], e1 = ev.e[1], e2 = ev.e[2];
    Launcher L = make_launcher(m, sh);
    L.prof = nullptr;
    int rc = FL_OK;
    auto one = [&](int f, int slot) -> int {
        if (f == 0) { FL_NCCL(ncclAllReduce(buf, buf, (size_t)n, ncclFloat, ncclSum, sh.comm, sh.stream)); return FL_OK; }
        if (f == 1) return oneshot(m, sh, false, buf, buf, n, 0);
        return launch_ll_allreduce(L, pc.ll_dev, slot, buf, buf, n);
    };
    const int slots = (int)(2 * m->D.L);
    for (int w = 0; w < 2 && rc == FL_OK; w++) {                 // w = 0: warm-up
        FL_HIP(hipEventRecord(e0, sh.stream));
        if (form == 2) {
            // a decode step's pattern: 2 L exchanges (slots 1 .. 2 L), then one one-shot collective moves the epoch
            for (int i = 0; i < iters && rc == FL_OK; i++) {
                for (int s2 = 1; s2 <= slots && rc == FL_OK; s2++) rc = one(2, s2);
                if (rc == FL_OK) rc = one(1, 0);
            }
            FL_HIP(hipEventRecord(e1, sh.stream));
            for (int i = 0; i < iters && rc == FL_OK; i++) rc = one(1, 0);
            FL_HIP(hipEventRecord(e2, sh.stream));
        } else {
            for (int i = 0; i < iters && rc == FL_OK; i++) rc = one(form, 0);
            FL_HIP(hipEventRecord(e1, sh.stream));
            FL_HIP(hipEventRecord(e2, sh.stream));
        }
        FL_HIP(hipStreamSynchronize(sh.stream));
    }
    float ms01 = 0.f, ms12 = 0.f;
    if (rc == FL_OK) { FL_HIP(hipEventElapsedTime(&ms01, e0, e1)); FL_HIP(hipEventElapsedTime(&ms12, e1, e2)); }
    FL_TRY(rc);
    if (pc.err && *pc.err) FL_FAIL(FL_ERR_RCCL, "comm_probe: a wait gave up (code 0x%x)", *pc.err);
    *us_per_call = form == 2 ? (double)(ms01 - ms12) * 1e3 / ((double)iters * slots) : (double)ms01 * 1e3 / iters;
    return FL_OK;
}

int comm_check(Model *m) {
    for (auto &sh : m->shards)
        if (sh.pc.err && *sh.pc.err) FL_FAIL(FL_ERR_RCCL, "one-shot collective gave up waiting for a peer (code 0x%x): the tensor-parallel group is broken", *sh.pc.err);
    return FL_OK;
}

// Decode step with the fused kernels: 5 launches per layer instead of 10.
//   K1 gemv[norm1(+residual add, or embedding) -> qkv -> RoPE -> q buffer / KV cache slot]
//   K2 attn_decode (+ in-launch split-S combine)      K3 gemv[o_proj] -> delta   (all-reduce)
//   K4 gemv[norm2(+add) -> gate/up -> silu*up]        K5 gemv[down]   -> delta   (all-reduce)
// and finally gemv[final norm -> lm_head].  The residual ping-pongs x_res <-> x_res2 because the
// norm prologue of every workgroup reads x_in while workgroup 0 writes the updated residual.
// In a tensor-parallel group whose LL regions are connected, K3 and K5 carry their all-reduce in the epilogue
// (comm_ll.h): the layer stays at 5 launches.
bool fused_all_reduce_ready(const Model *m) {
    // 0: never; 1: when every rank has a GPU of its own (two full-chip GEMV grids that wait for each other's rows
    // cannot both be resident on one card -- the same-device rehearsals); 2: regardless (tests with small grids)
    const int allow = tune(TK_TP_FUSED_AR);
    if (!allow || !m->fused_decode) return false;
    if (allow < 2) for (auto &sh : m->shards) if (sh.pc.shares_device) return false;
    const bool group_of_one = m->tp == 1 && m->shards[0].pc.connected;          // FL_DEBUG_RCCL_SELF: the bootstrap rehearsal
    if (m->tp_mode != FL_TP_MULTI_PROCESS && m->tp_mode != FL_TP_SINGLE_PROCESS && !group_of_one) return false;   // emulated shards share one stream
    for (auto &sh : m->shards) {
        if (!sh.pc.connected || !sh.pc.ll_ok || sh.Vs > sh.pc.nmax) return false;   // the per-step logits gather must move the epoch counter
        if (!gemv_supported(m->dtype, m->D.h, sh.Hs * m->D.d) || !gemv_supported(m->dtype, m->D.h, sh.Ip)) return false;
    }
    return true;
}

}  // namespace fl
