// engine.cpp — demucs_inference over several GPUs of one node and over a bag of models, in one process.
//
// Replaces, for a caller with more than one MI355X (or more than one model), the loop nest of
//   cli-apps/demucs_ft.cpp:221-241   (four demucs_inference calls, stem i from model i)
//   src/model_apply.cpp:189-235      (for each overlapping segment: segment_inference)
// (/root/reference). That nest is embarrassingly parallel: every (model m, segment i) work item reads a
// slice of the shifted track and produces an independent (S, 2, segment) block; the only coupling is
// the weighted overlap-add of one model's blocks. So:
//
//   * work items are enumerated model-major, (m, 0..n_seg(m)-1) with n_seg(m) from model m's own shift
//     offset (model_apply.cpp:114), T items in total;
//   * logical device l owns the CONTIGUOUS item range [l*T/G, (l+1)*T/G): at most two models per device
//     for the bag of 4 on 8 GPUs (21 items = one batch each), one contiguous slab of results per
//     (device, model) run, and every item costs the same (short tails are zero-padded, Q8);
//   * each device runs on its own host thread, stream and context (weights replicated: 170 MB fp32 per
//     model), batches of <= max_batch items through dmx_segment_infer_device;
//   * ONE exchange step: every (device, model) slab is gathered into the root device's
//     [n_seg][S][2][segment] buffer of that model -
//         transport RCCL : ncclSend on the owner / grouped ncclRecv on the root, one communicator per
//                          device from ncclCommInitAll (xGMI; 11 MB per 4-source segment);
//         transport P2P  : hipMemcpyPeerAsync pushed by the owner (SDMA over xGMI, no CU on either side);
//                          also the only transport when one HIP device backs several logical devices
//                          (tests on a 1-GPU box), where RCCL cannot build a communicator;
//   * the root overlap-adds every model's blocks in segment order (bit-identical to the 1-device result:
//     each output sample accumulates its <= 2 covering segments in increasing index), takes stem m from
//     model m for a bag, de-normalises and copies out.
//
// Optional finish mode OWNER (dmx_engine_set_finish / env DMX_FINISH=owner): instead of gathering whole segment
// blocks on the root, the owner of the contiguous run [g0, g1) also FINISHES the output stretch
// [g0*stride, g1*stride) of the shifted track: the only data it lacks is the tail [stride, segment) of
// segment g0-1 (S*2*(segment-stride) floats = 2.75 MB against 11 MB per gathered block and 231 MB per
// gathered 4-minute track), which the previous owner sends (ncclSend/ncclRecv or a peer copy); every
// device overlap-adds and copies out its own stretch. Same accumulation order per sample, same bits.
//
// librccl is bound lazily with dlopen the first time an RCCL engine is created: single-device users and
// torch.distributed processes (which bring their own copy) never load a second RCCL.
#include "api_internal.h"

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <mutex>
#include <thread>

using namespace dmx;

// ---- the part of the RCCL API used here (rccl.h), bound at run time
typedef struct ncclComm *ncclComm_t;
typedef int ncclResult_t; // ncclSuccess == 0
static const int kNcclFloat = 7; // ncclFloat32 (rccl.h: ncclInt8 0, ..., ncclFloat16 6, ncclFloat32 7)
struct RcclApi
{
    void *handle = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string why;
};
static RcclApi *rccl_api()
{
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
            if ((api.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL)))
                break;
        if (!api.handle)
        {
            api.why = std::string("cannot load librccl: ") + dlerror();
            return;
        }
        auto sym = [&](const char *n) {
            void *p = dlsym(api.handle, n);
            if (!p)
                api.why += std::string(" missing symbol ") + n;
            return p;
        };
        api.CommInitAll = (decltype(api.CommInitAll))sym("ncclCommInitAll");
        api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
        api.Send = (decltype(api.Send))sym("ncclSend");
        api.Recv = (decltype(api.Recv))sym("ncclRecv");
        api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
        api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
        api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
    });
    return &api;
}
#define NCCLCHK(expr)                                                                                     \
    do                                                                                                    \
    {                                                                                                     \
        ncclResult_t r_ = (expr);                                                                         \
        if (r_ != 0)                                                                                      \
            return dmx_fail(DMX_ERR_HIP, "%s failed: %s", #expr, rccl_api()->GetErrorString(r_));           \
    } while (0)

struct EngineDev
{
    int dev = 0;                     // HIP device id (several logical devices may share one)
    std::vector<dmx_model *> models; // replica of every model on this device
    dmx_ctx *ctx = nullptr;          // one arena, rebound to the model of the run (dmx_ctx_set_model)
    DevBuf slab;                     // results of this device's items (non-root devices; every device in OWNER mode)
    DevBuf haloSend, haloRecv;       // OWNER mode: packed tails [stride, seg) of boundary segments, one slot per run
    float *pinned = nullptr;         // OWNER mode: pinned staging of this device's finished stretches
    i64 pinnedCap = 0;
    ncclComm_t comm = nullptr;
    hipEvent_t evDone = nullptr;
    // result of the last call's worker
    int rc = DMX_OK;
    std::string err;
};

struct dmx_engine
{
    int nModels = 0, S = 0, maxBatch = 0, transport = DMX_TRANSPORT_P2P, finish = DMX_FINISH_ROOT;
    bool rcclSelf = false;      // test hook DMX_RCCL_SELF=1: several logical devices on ONE GPU exchange their slabs through
                                // ncclSend / ncclRecv to self on a 1-rank communicator (posted by the root worker)
    hipEvent_t evExchange = nullptr; // rcclSelf + OWNER: the self exchange has been enqueued on the root's stream
    i64 seg = 0;
    std::vector<EngineDev> devs;
    std::vector<DevBuf> segOut; // root: per model [n_seg][S][2][seg]
    DevBuf out;                 // root: (S, 2, n)
    std::mutex mu;              // one track at a time per engine: demucs_inference is called concurrently on a
                                // shared const model by the reference's threaded driver (threaded_inference.hpp:105-123)
    ~dmx_engine()
    {
        RcclApi *api = transport == DMX_TRANSPORT_RCCL ? rccl_api() : nullptr;
        for (EngineDev &d : devs)
        {
            (void)hipSetDevice(d.dev);
            if (d.comm && api && api->CommDestroy && (!rcclSelf || &d == &devs[0]))
                (void)api->CommDestroy(d.comm);
            if (d.evDone)
                (void)hipEventDestroy(d.evDone);
            delete d.ctx; // before the models it points to
            if (d.pinned)
                (void)hipHostFree(d.pinned);
            for (DevBuf *b : {&d.slab, &d.haloSend, &d.haloRecv})
                if (b->p)
                    (void)hipFree(b->p);
            for (dmx_model *m : d.models)
                dmx_model_free(m);
        }
        if (!devs.empty())
        {
            (void)hipSetDevice(devs[0].dev);
            if (evExchange)
                (void)hipEventDestroy(evExchange);
            for (DevBuf &b : segOut)
                if (b.p)
                    (void)hipFree(b.p);
            if (out.p)
                (void)hipFree(out.p);
        }
    }
};

extern "C" int dmx_engine_create(const char *const *model_files, int n_models, const int *devices, int n_devices, int max_batch,
                                 int transport, dmx_engine **out)
{
    if (!model_files || !out || n_models < 1 || n_models > 16 || (n_devices > 0 && !devices) || n_devices > 64 || max_batch < 1 ||
        max_batch > 64)
        return dmx_fail(DMX_ERR_ARG, "dmx_engine_create: invalid argument");
    *out = nullptr;
    const int visible = dmx_device_count();
    if (visible <= 0)
    {
        // same order of failures as dmx_model_load: an unreadable / malformed file is reported as such,
        // a good file fails with DMX_ERR_NO_DEVICE (there is no CPU fallback)
        dmx_model *h = nullptr;
        const int rc = dmx_model_load(model_files[0], 0, &h);
        return rc != DMX_OK ? rc : dmx_fail(DMX_ERR_NO_DEVICE, "dmx_engine_create: no HIP device available");
    }
    auto e = std::make_unique<dmx_engine>();
    e->nModels = n_models;
    e->maxBatch = max_batch;
    std::vector<int> devs;
    if (n_devices <= 0) // all visible devices
        for (int i = 0; i < visible; ++i)
            devs.push_back(i);
    else
        devs.assign(devices, devices + n_devices);
    bool distinct = true;
    for (size_t i = 0; i < devs.size(); ++i)
    {
        if (devs[i] < 0 || devs[i] >= visible)
            return dmx_fail(DMX_ERR_ARG, "dmx_engine_create: device %d out of range (have %d)", devs[i], visible);
        for (size_t j = 0; j < i; ++j)
            distinct = distinct && devs[j] != devs[i];
    }
    if (transport == DMX_TRANSPORT_AUTO)
    {
        const char *env = getenv("DMX_GATHER");
        if (env && !strcmp(env, "p2p"))
            transport = DMX_TRANSPORT_P2P;
        else if (env && !strcmp(env, "rccl"))
            transport = DMX_TRANSPORT_RCCL;
        else
        {
            transport = (distinct && devs.size() > 1) ? DMX_TRANSPORT_RCCL : DMX_TRANSPORT_P2P;
            // AUTO never fails for want of librccl: the peer-copy transport moves the same bytes over the same links
            if (transport == DMX_TRANSPORT_RCCL)
            {
                RcclApi *api = rccl_api();
                if (!api->handle || !api->why.empty())
                    transport = DMX_TRANSPORT_P2P;
            }
        }
    }
    if (transport != DMX_TRANSPORT_P2P && transport != DMX_TRANSPORT_RCCL)
        return dmx_fail(DMX_ERR_ARG, "dmx_engine_create: unknown transport %d", transport);
    if (transport == DMX_TRANSPORT_RCCL && !distinct)
    {
        // Test hook for 1-GPU boxes: all logical devices on ONE GPU, one 1-rank communicator; every slab / tail travels
        // through ncclSend + ncclRecv to self, so the RCCL data path (argument order, counts, offsets, stream ordering)
        // is executed on hardware even where no second GPU exists.
        bool same = true;
        for (int v : devs)
            same = same && v == devs[0];
        const char *self = getenv("DMX_RCCL_SELF");
        if (!(self && !strcmp(self, "1") && same))
            return dmx_fail(DMX_ERR_ARG, "dmx_engine_create: the RCCL transport needs distinct devices (a communicator holds a GPU once)");
        e->rcclSelf = true;
    }
    e->transport = transport;
    e->devs.resize(devs.size());
    for (size_t l = 0; l < devs.size(); ++l)
    {
        EngineDev &d = e->devs[l];
        d.dev = devs[l];
        for (int m = 0; m < n_models; ++m)
        {
            dmx_model *h = nullptr;
            if (l == 0) // parse + pack once per file; the other devices get a copy of the packed weights
                DMXCHK(dmx_model_load(model_files[m], d.dev, &h));
            else
                DMXCHK(dmx_model_clone(e->devs[0].models[(size_t)m], d.dev, &h));
            d.models.push_back(h);
            if (dmx_model_n_sources(h) != dmx_model_n_sources(d.models[0]) || dmx_model_arch(h) != dmx_model_arch(d.models[0]))
                return dmx_fail(DMX_ERR_ARG, "dmx_engine_create: the models of a bag must have the same architecture and number of sources");
        }
        DMXCHK(dmx_ctx_create(d.models[0], 0, max_batch, &d.ctx));
        for (int m = 1; m < n_models; ++m) // every model must fit the one plan
        {
            DMXCHK(dmx_ctx_set_model(d.ctx, d.models[(size_t)m]));
        }
        DMXCHK(dmx_ctx_set_model(d.ctx, d.models[0]));
        HIPCHK(hipSetDevice(d.dev));
        HIPCHK(hipEventCreateWithFlags(&d.evDone, hipEventDisableTiming));
    }
    e->S = dmx_model_n_sources(e->devs[0].models[0]);
    e->seg = e->devs[0].ctx->seg;
    if (n_models > 1 && n_models != e->S)
        return dmx_fail(DMX_ERR_ARG, "dmx_engine_create: a bag needs one model per source (%d models, %d sources): stem i is taken from model i",
                        n_models, e->S);
    e->segOut.resize((size_t)n_models);
    if (devs.size() > 1 && distinct)
        for (size_t l = 0; l < devs.size(); ++l) // direct xGMI copies: every device to the root (gather) and to
            for (size_t q = 0; q < devs.size(); ++q) // the later devices (halo of the OWNER finish mode)
            {
                if (q == l || (q != 0 && q < l))
                    continue;
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, devs[l], devs[q]) == hipSuccess && can)
                {
                    (void)hipSetDevice(devs[l]);
                    hipError_t pe = hipDeviceEnablePeerAccess(devs[q], 0);
                    if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled)
                        return dmx_fail(DMX_ERR_HIP, "hipDeviceEnablePeerAccess(%d -> %d) failed: %s", devs[l], devs[q],
                                        hipGetErrorString(pe));
                    (void)hipGetLastError();
                }
            }
    if (const char *env = getenv("DMX_FINISH"))
    {
        if (!strcmp(env, "owner"))
            e->finish = DMX_FINISH_OWNER;
        else if (strcmp(env, "root"))
            return dmx_fail(DMX_ERR_ARG, "DMX_FINISH must be root or owner (is \"%s\")", env);
    }
    if (transport == DMX_TRANSPORT_RCCL)
    {
        RcclApi *api = rccl_api();
        if (!api->handle || !api->why.empty())
            return dmx_fail(DMX_ERR_HIP, "dmx_engine_create: RCCL transport unavailable (%s); DMX_GATHER=p2p selects peer copies", api->why.c_str());
        if (e->rcclSelf)
        {
            ncclComm_t comm = nullptr;
            NCCLCHK(api->CommInitAll(&comm, 1, devs.data()));
            for (size_t l = 0; l < devs.size(); ++l)
                e->devs[l].comm = comm;
            HIPCHK(hipSetDevice(devs[0]));
            HIPCHK(hipEventCreateWithFlags(&e->evExchange, hipEventDisableTiming));
        }
        else
        {
            std::vector<ncclComm_t> comms(devs.size(), nullptr);
            NCCLCHK(api->CommInitAll(comms.data(), (int)devs.size(), devs.data()));
            for (size_t l = 0; l < devs.size(); ++l)
                e->devs[l].comm = comms[l];
        }
    }
    *out = e.release();
    return DMX_OK;
}

extern "C" void dmx_engine_free(dmx_engine *e) { delete e; }
extern "C" int dmx_engine_n_devices(const dmx_engine *e) { return e ? (int)e->devs.size() : 0; }
extern "C" int dmx_engine_n_models(const dmx_engine *e) { return e ? e->nModels : 0; }
extern "C" int dmx_engine_n_sources(const dmx_engine *e) { return e ? e->S : 0; }
extern "C" int dmx_engine_arch(const dmx_engine *e) { return e && !e->devs.empty() ? dmx_model_arch(e->devs[0].models[0]) : 0; }
extern "C" int dmx_engine_transport(const dmx_engine *e) { return e ? e->transport : -1; }
extern "C" int dmx_engine_finish(const dmx_engine *e) { return e ? e->finish : -1; }
extern "C" int dmx_engine_set_finish(dmx_engine *e, int finish)
{
    if (!e || (finish != DMX_FINISH_ROOT && finish != DMX_FINISH_OWNER))
        return dmx_fail(DMX_ERR_ARG, "dmx_engine_set_finish: invalid argument");
    std::lock_guard<std::mutex> guard(e->mu);
    e->finish = finish;
    return DMX_OK;
}
extern "C" dmx_ctx *dmx_engine_root_ctx(dmx_engine *e, int model)
{
    if (!e || model < 0 || model >= e->nModels)
        return nullptr;
    // the rebind is ordered against a running dmx_engine_track_infer; the CALLS the caller then makes on the returned
    // context are not: a context is single-threaded (include/demucs_hip.h), the C++ shim serialises them with the model lock
    std::lock_guard<std::mutex> guard(e->mu);
    if (dmx_ctx_set_model(e->devs[0].ctx, e->devs[0].models[(size_t)model]) != DMX_OK)
        return nullptr;
    return e->devs[0].ctx;
}

namespace
{
struct Run // a maximal stretch of one device's items that belongs to one model
{
    int model, g0, g1;  // segments [g0, g1) of `model`
    i64 slabOff;        // offset (in segment blocks) of the run's results in the device's slab
    // OWNER finish mode
    i64 ownSlot = 0;            // block index of segment g0 in the slab (the block before it is the halo when g0 > 0)
    int predDev = -1, predRun = -1; // who computes segment g0-1 of this model
    int succDev = -1, succRun = -1; // who computes segment g1
};
struct Shared
{
    std::mutex mu;
    std::condition_variable cv;
    int itemsDone = 0;
    int workersLeft = 0;
    std::vector<std::string> messages; // progress lines waiting for the calling thread
    // OWNER finish mode, peer-copy transport: haloReady[l][ri] is set by the sender once its copy has landed
    std::vector<std::vector<char>> haloReady;
    bool failed = false; // a worker gave up: receivers stop waiting
    int arrived[2] = {0, 0}; // host barriers of the RCCL transports (agree())
};
} // namespace

// test hook: DMX_TEST_FAIL_DEV=<logical device> makes that device's compute phase fail (the exchange must then be
// skipped by everybody instead of hanging: tests/test_gpu_parity.py)
static bool fault_injected(int l)
{
    const char *v = getenv("DMX_TEST_FAIL_DEV");
    return v && *v && atoi(v) == l;
}

// Host barrier of the RCCL transports. A collective-style exchange deadlocks when one participant never posts its
// half (its peers sit in hipStreamSynchronize behind an unmatched ncclRecv / ncclSend), so the workers first AGREE
// that every one of them got through its compute phase: each arrives exactly once per barrier `which` with its own
// status; the call returns true only if all G arrived healthy. Nobody posts an RCCL call otherwise.
static bool agree(Shared &sh, int which, int G, bool ok)
{
    std::unique_lock<std::mutex> lk(sh.mu);
    if (!ok)
        sh.failed = true;
    ++sh.arrived[which];
    sh.cv.notify_all();
    sh.cv.wait(lk, [&] { return sh.arrived[which] >= G; });
    return !sh.failed;
}

// items of model m: first global item index
static void partition(const std::vector<int> &nseg, int G, std::vector<std::vector<Run>> &runs)
{
    int T = 0;
    for (int v : nseg)
        T += v;
    runs.assign((size_t)G, {});
    for (int l = 0; l < G; ++l)
    {
        const int lo = (int)((i64)l * T / G), hi = (int)((i64)(l + 1) * T / G);
        int base = 0;
        i64 off = 0;
        for (int m = 0; m < (int)nseg.size(); ++m)
        {
            const int a = std::max(lo, base), b = std::min(hi, base + nseg[(size_t)m]);
            if (b > a)
            {
                runs[(size_t)l].push_back(Run{m, a - base, b - base, off});
                off += b - a;
            }
            base += nseg[(size_t)m];
        }
    }
}

// pure host function, exported so that the dealing can be checked without a GPU (tests/test_engine_cpu.py):
// runs[(l*n_models + m)*2 + {0,1}] = segment range [g0, g1) of model m owned by logical device l
extern "C" int dmx_engine_partition(const int *n_segments, int n_models, int n_devices, int *runs_out)
{
    if (!n_segments || !runs_out || n_models < 1 || n_devices < 1)
        return dmx_fail(DMX_ERR_ARG, "dmx_engine_partition: invalid argument");
    std::vector<int> nseg(n_segments, n_segments + n_models);
    std::vector<std::vector<Run>> runs;
    partition(nseg, n_devices, runs);
    for (int i = 0; i < n_devices * n_models * 2; ++i)
        runs_out[i] = 0;
    for (int l = 0; l < n_devices; ++l)
        for (const Run &r : runs[(size_t)l])
        {
            runs_out[(l * n_models + r.model) * 2] = r.g0;
            runs_out[(l * n_models + r.model) * 2 + 1] = r.g1;
        }
    return DMX_OK;
}

// upload of the track + its statistics (every device holds the small track and computes the same numbers)
static int upload_track(dmx_ctx *c, const float *audio, int layout, i64 n)
{
    if (layout == DMX_LAYOUT_EIGEN)
        HIPCHK(hipMemcpyAsync(c->bAudio.p, audio, sizeof(float) * 2 * (size_t)n, hipMemcpyHostToDevice, c->stream));
    else
    {
        HIPCHK(hipMemcpyAsync(c->bTmp.p, audio, sizeof(float) * 2 * (size_t)n, hipMemcpyHostToDevice, c->stream));
        launch_planar_to_interleaved(c->bTmp.p, c->bAudio.p, n, c->stream);
    }
    return dmx_track_stats_device(c, c->bAudio.p, n, c->dStats);
}

static void post_progress(Shared &sh, int items, int l, int dev)
{
    std::lock_guard<std::mutex> lk(sh.mu);
    sh.itemsDone += items;
    char msg[160];
    snprintf(msg, sizeof(msg), "2., apply model w/ split, device %d (gpu %d) finished %d segments", l, dev, items);
    sh.messages.push_back(msg);
    sh.cv.notify_all();
}

// what logical device l does for one track (on its own host thread)
static int device_work(dmx_engine *e, int l, const float *audio, int layout, i64 n, const std::vector<int> &shifts,
                       const std::vector<int> &nseg, const std::vector<Run> &runs, const std::vector<std::vector<Run>> &allRuns,
                       Shared &sh)
{
    EngineDev &d = e->devs[(size_t)l];
    dmx_ctx *c = d.ctx;
    const int S = e->S;
    const i64 seg = e->seg, blk = (i64)S * 2 * seg;
    std::vector<int> idx;
    size_t nev = 0;
    std::vector<int> evItems;
    const bool rccl = e->transport == DMX_TRANSPORT_RCCL && e->devs.size() > 1;
    auto compute = [&]() -> int {
        HIPCHK(hipSetDevice(d.dev));
        if (fault_injected(l))
            return dmx_fail(DMX_ERR_HIP, "device %d: injected fault (DMX_TEST_FAIL_DEV)", l);
        if (!runs.empty() || l == 0)
        {
            DMXCHK(upload_track(c, audio, layout, n));
        }
        for (const Run &r : runs)
        {
            DMXCHK(dmx_ctx_set_model(c, d.models[(size_t)r.model]));
            float *dst = l == 0 ? e->segOut[(size_t)r.model].p + (i64)r.g0 * blk : d.slab.p + r.slabOff * blk;
            for (int g = r.g0; g < r.g1; g += e->maxBatch)
            {
                const int nb = std::min(e->maxBatch, r.g1 - g);
                idx.resize((size_t)nb);
                for (int i = 0; i < nb; ++i)
                    idx[(size_t)i] = g + i;
                DMXCHK(dmx_track_gather_device(c, c->bAudio.p, n, c->dStats, shifts[(size_t)r.model], idx.data(), nb, c->bMix.p));
                DMXCHK(dmx_segment_infer_device(c, c->bMix.p, dst + (i64)(g - r.g0) * blk, nb));
                hipEvent_t ev = dmx_batch_event(c, nev++);
                if (!ev)
                    return dmx_fail(DMX_ERR_HIP, "hipEventCreate failed");
                HIPCHK(hipEventRecord(ev, c->stream));
                evItems.push_back(nb);
            }
        }
        if (e->rcclSelf)
            HIPCHK(hipEventRecord(d.evDone, c->stream)); // the root's stream waits for it before the self exchange
        return DMX_OK;
    };
    const int crc = compute();
    if (rccl)
    {
        // no worker posts an RCCL call unless every worker got here healthy (a missing partner would hang the rest)
        const std::string cerr = crc == DMX_OK ? std::string() : dmx_err_string();
        if (!agree(sh, 0, (int)e->devs.size(), crc == DMX_OK))
        {
            if (crc != DMX_OK)
            {
                dmx_set_err_string(cerr);
                return crc;
            }
            return dmx_fail(DMX_ERR_HIP, "device %d: another device failed before the exchange; no RCCL call was posted", l);
        }
    }
    else if (crc != DMX_OK)
        return crc;
    // ---- the exchange step
    if (rccl && e->rcclSelf)
    {
        // one GPU, one 1-rank communicator: the root posts every slab as ncclSend + ncclRecv to self in ONE group
        RcclApi *api = rccl_api();
        if (l == 0)
        {
            for (size_t p = 1; p < allRuns.size(); ++p)
                HIPCHK(hipStreamWaitEvent(c->stream, e->devs[p].evDone, 0));
            NCCLCHK(api->GroupStart());
            for (size_t p = 1; p < allRuns.size(); ++p)
                for (const Run &r : allRuns[p])
                {
                    const size_t cnt = (size_t)((i64)(r.g1 - r.g0) * blk);
                    NCCLCHK(api->Send(e->devs[p].slab.p + r.slabOff * blk, cnt, kNcclFloat, 0, d.comm, c->stream));
                    NCCLCHK(api->Recv(e->segOut[(size_t)r.model].p + (i64)r.g0 * blk, cnt, kNcclFloat, 0, d.comm, c->stream));
                }
            NCCLCHK(api->GroupEnd());
        }
    }
    else if (rccl)
    {
        RcclApi *api = rccl_api();
        if (l != 0)
        {
            for (const Run &r : runs) // one message per (device, model) run, in run order
                NCCLCHK(api->Send(d.slab.p + r.slabOff * blk, (size_t)((i64)(r.g1 - r.g0) * blk), kNcclFloat, 0, d.comm, c->stream));
        }
        else
        {
            NCCLCHK(api->GroupStart());
            for (size_t p = 1; p < allRuns.size(); ++p)
                for (const Run &r : allRuns[p])
                    NCCLCHK(api->Recv(e->segOut[(size_t)r.model].p + (i64)r.g0 * blk, (size_t)((i64)(r.g1 - r.g0) * blk), kNcclFloat, (int)p,
                                      d.comm, c->stream));
            NCCLCHK(api->GroupEnd());
        }
    }
    else if (l != 0)
    {
        const int rootDev = e->devs[0].dev;
        for (const Run &r : runs)
            HIPCHK(hipMemcpyPeerAsync(e->segOut[(size_t)r.model].p + (i64)r.g0 * blk, rootDev, d.slab.p + r.slabOff * blk, d.dev,
                                      sizeof(float) * (size_t)((i64)(r.g1 - r.g0) * blk), c->stream));
    }
    (void)nseg;
    // ---- progress: one line per finished batch, delivered by the calling thread
    for (size_t k = 0; k < nev; ++k)
    {
        HIPCHK(hipEventSynchronize(c->batchEvents[k]));
        post_progress(sh, evItems[k], l, d.dev);
    }
    DMXCHK(dmx_ctx_sync_checked(c)); // the slab has reached the root (or the root has received every slab)
    return DMX_OK;
}

// OWNER finish mode: what logical device l does for one track. The device computes its runs, sends the
// tail of each run's last segment to the owner of the next segment, receives the tail in front of each
// of its runs, overlap-adds the stretch [g0*stride, g1*stride) of every run and copies it into `out`.
static int device_work_owner(dmx_engine *e, int l, const float *audio, int layout, i64 n, const std::vector<int> &shifts,
                             const std::vector<int> &nseg, const std::vector<i64> &lens, i64 stride,
                             const std::vector<std::vector<Run>> &allRuns, float *out, Shared &sh)
{
    EngineDev &d = e->devs[(size_t)l];
    dmx_ctx *c = d.ctx;
    const std::vector<Run> &runs = allRuns[(size_t)l];
    const int S = e->S, M = e->nModels;
    const i64 seg = e->seg, blk = (i64)S * 2 * seg, tail = seg - stride, halo = (i64)S * 2 * tail;
    const bool rccl = e->transport == DMX_TRANSPORT_RCCL;
    const int G = (int)e->devs.size();
    std::vector<int> idx, evItems;
    size_t nev = 0;
    // compute phase + packing of the tails [stride, seg) of the boundary segments (S*2 rows of `tail` floats)
    auto compute = [&]() -> int {
        if (fault_injected(l))
            return dmx_fail(DMX_ERR_HIP, "device %d: injected fault (DMX_TEST_FAIL_DEV)", l);
        if (runs.empty())
            return DMX_OK;
        HIPCHK(hipSetDevice(d.dev));
        DMXCHK(upload_track(c, audio, layout, n));
        for (const Run &r : runs)
        {
            DMXCHK(dmx_ctx_set_model(c, d.models[(size_t)r.model]));
            float *dst = d.slab.p + r.ownSlot * blk;
            for (int g = r.g0; g < r.g1; g += e->maxBatch)
            {
                const int nb = std::min(e->maxBatch, r.g1 - g);
                idx.resize((size_t)nb);
                for (int i = 0; i < nb; ++i)
                    idx[(size_t)i] = g + i;
                DMXCHK(dmx_track_gather_device(c, c->bAudio.p, n, c->dStats, shifts[(size_t)r.model], idx.data(), nb, c->bMix.p));
                DMXCHK(dmx_segment_infer_device(c, c->bMix.p, dst + (i64)(g - r.g0) * blk, nb));
                hipEvent_t ev = dmx_batch_event(c, nev++);
                if (!ev)
                    return dmx_fail(DMX_ERR_HIP, "hipEventCreate failed");
                HIPCHK(hipEventRecord(ev, c->stream));
                evItems.push_back(nb);
            }
        }
        for (size_t ri = 0; ri < runs.size(); ++ri)
            if (runs[ri].succDev >= 0)
                launch_copy_rows(d.haloSend.p + (i64)ri * halo, tail, d.slab.p + (runs[ri].ownSlot + (runs[ri].g1 - 1 - runs[ri].g0)) * blk + stride,
                                 seg, tail, S * 2, c->stream);
        HIPCHK(hipGetLastError());
        if (e->rcclSelf)
            HIPCHK(hipEventRecord(d.evDone, c->stream));
        return DMX_OK;
    };
    const int crc = compute();
    if (rccl)
    {
        // every worker (also one without items) arrives: nobody posts an RCCL call unless all are healthy
        const std::string cerr = crc == DMX_OK ? std::string() : dmx_err_string();
        if (!agree(sh, 0, G, crc == DMX_OK))
        {
            if (crc != DMX_OK)
            {
                dmx_set_err_string(cerr);
                return crc;
            }
            return dmx_fail(DMX_ERR_HIP, "device %d: another device failed before the exchange; no RCCL call was posted", l);
        }
    }
    else if (crc != DMX_OK)
        return crc;
    if (rccl && e->rcclSelf)
    {
        // one GPU, one 1-rank communicator: the root posts every tail as ncclSend + ncclRecv to self in ONE group on
        // its stream, then every worker orders its own stream behind that exchange
        RcclApi *api = rccl_api();
        int xrc = DMX_OK;
        std::string xerr;
        if (l == 0)
        {
            auto post = [&]() -> int {
                dmx_ctx *c0 = e->devs[0].ctx;
                HIPCHK(hipSetDevice(e->devs[0].dev));
                for (int q = 0; q < G; ++q)
                    if (!allRuns[(size_t)q].empty())
                        HIPCHK(hipStreamWaitEvent(c0->stream, e->devs[(size_t)q].evDone, 0));
                NCCLCHK(api->GroupStart());
                for (int q = 0; q < G; ++q)
                    for (size_t ri = 0; ri < allRuns[(size_t)q].size(); ++ri)
                    {
                        const Run &r = allRuns[(size_t)q][ri];
                        if (r.succDev < 0)
                            continue;
                        EngineDev &dst = e->devs[(size_t)r.succDev];
                        NCCLCHK(api->Send(e->devs[(size_t)q].haloSend.p + (i64)ri * halo, (size_t)halo, kNcclFloat, 0, d.comm, c0->stream));
                        NCCLCHK(api->Recv(dst.haloRecv.p + (i64)r.succRun * halo, (size_t)halo, kNcclFloat, 0, d.comm, c0->stream));
                    }
                NCCLCHK(api->GroupEnd());
                HIPCHK(hipEventRecord(e->evExchange, c0->stream));
                return DMX_OK;
            };
            xrc = post();
            if (xrc != DMX_OK)
                xerr = dmx_err_string();
        }
        if (!agree(sh, 1, G, xrc == DMX_OK)) // the exchange event exists (recorded) before anyone waits on it
        {
            if (xrc != DMX_OK)
            {
                dmx_set_err_string(xerr);
                return xrc;
            }
            return dmx_fail(DMX_ERR_HIP, "device %d: the self exchange could not be posted", l);
        }
        if (runs.empty())
            return DMX_OK;
        HIPCHK(hipStreamWaitEvent(c->stream, e->evExchange, 0));
    }
    else if (runs.empty())
        return DMX_OK;
    else if (rccl)
    {
        RcclApi *api = rccl_api();
        NCCLCHK(api->GroupStart());
        for (size_t ri = 0; ri < runs.size(); ++ri)
        {
            if (runs[ri].succDev >= 0)
                NCCLCHK(api->Send(d.haloSend.p + (i64)ri * halo, (size_t)halo, kNcclFloat, runs[ri].succDev, d.comm, c->stream));
            if (runs[ri].predDev >= 0)
                NCCLCHK(api->Recv(d.haloRecv.p + (i64)ri * halo, (size_t)halo, kNcclFloat, runs[ri].predDev, d.comm, c->stream));
        }
        NCCLCHK(api->GroupEnd());
    }
    else
    {
        for (size_t ri = 0; ri < runs.size(); ++ri)
            if (runs[ri].succDev >= 0)
            {
                EngineDev &q = e->devs[(size_t)runs[ri].succDev];
                HIPCHK(hipMemcpyPeerAsync(q.haloRecv.p + (i64)runs[ri].succRun * halo, q.dev, d.haloSend.p + (i64)ri * halo, d.dev,
                                          sizeof(float) * (size_t)halo, c->stream));
            }
    }
    // ---- progress: one line per finished batch, delivered by the calling thread
    for (size_t k = 0; k < nev; ++k)
    {
        HIPCHK(hipEventSynchronize(c->batchEvents[k]));
        post_progress(sh, evItems[k], l, d.dev);
    }
    if (!rccl)
    {
        DMXCHK(dmx_ctx_sync_checked(c)); // my tails have landed at their receivers
        {
            std::lock_guard<std::mutex> lk(sh.mu);
            for (const Run &r : runs)
                if (r.succDev >= 0)
                    sh.haloReady[(size_t)r.succDev][(size_t)r.succRun] = 1;
            sh.cv.notify_all();
        }
        std::unique_lock<std::mutex> lk(sh.mu);
        for (size_t ri = 0; ri < runs.size(); ++ri)
            if (runs[ri].predDev >= 0)
            {
                sh.cv.wait(lk, [&] { return sh.haloReady[(size_t)l][ri] || sh.failed; });
                if (!sh.haloReady[(size_t)l][ri])
                    return dmx_fail(DMX_ERR_HIP, "device %d: the owner of the previous segment failed", l);
            }
    }
    // ---- finish my stretches
    struct Piece
    {
        i64 stage, dst, len, pitch;
        int rows;
    };
    std::vector<Piece> pieces;
    i64 stageFloats = 0;
    for (size_t ri = 0; ri < runs.size(); ++ri)
    {
        const Run &r = runs[ri];
        const int m = r.model;
        if (r.predDev >= 0) // unpack the received tail into the block in front of the run (segment g0-1)
            launch_copy_rows(d.slab.p + (r.ownSlot - 1) * blk + stride, seg, d.haloRecv.p + (i64)ri * halo, tail, tail, S * 2, c->stream);
        // positions [g0*stride, g1*stride) of the shifted track (to its end for the last run), as samples of the result
        const i64 off = DMX_MAX_SHIFT - shifts[(size_t)m];
        const i64 j1 = r.g1 == nseg[(size_t)m] ? lens[(size_t)m] : (i64)r.g1 * stride;
        const i64 i0 = std::min(n, std::max<i64>(0, (i64)r.g0 * stride - off)), i1 = std::min(n, std::max<i64>(0, j1 - off));
        if (i1 <= i0)
            continue;
        const int gBase = r.predDev >= 0 ? r.g0 - 1 : r.g0;
        const int p0 = M == 1 ? 0 : 2 * m, np = M == 1 ? 2 * S : 2;
        launch_track_ola(d.slab.p + (r.ownSlot - (r.g0 - gBase)) * blk, nseg[(size_t)m], S, seg, stride, lens[(size_t)m], n,
                         shifts[(size_t)m], c->dStats, c->bOut.p, layout == DMX_LAYOUT_EIGEN ? 1 : 0, p0, np, i0, i1, c->stream, gBase);
        HIPCHK(hipGetLastError());
        // D2H into this device's pinned staging buffer (G devices write one pageable result concurrently: each
        // through its own DMA into its own pinned pages, then a host copy into the disjoint range it owns)
        Piece pc;
        pc.stage = stageFloats;
        if (layout == DMX_LAYOUT_EIGEN) // (only with M == 1: all planes of the samples [i0, i1) are contiguous)
        {
            pc.dst = (i64)2 * S * i0, pc.len = (i64)2 * S * (i1 - i0), pc.rows = 1, pc.pitch = 0;
            HIPCHK(hipMemcpyAsync(d.pinned + pc.stage, c->bOut.p + pc.dst, sizeof(float) * (size_t)pc.len, hipMemcpyDeviceToHost, c->stream));
        }
        else
        {
            pc.dst = (i64)p0 * n + i0, pc.len = i1 - i0, pc.rows = np, pc.pitch = n;
            for (int pl = 0; pl < np; ++pl)
                HIPCHK(hipMemcpyAsync(d.pinned + pc.stage + (i64)pl * pc.len, c->bOut.p + pc.dst + (i64)pl * n, sizeof(float) * (size_t)pc.len,
                                      hipMemcpyDeviceToHost, c->stream));
        }
        stageFloats += pc.len * pc.rows;
        pieces.push_back(pc);
    }
    DMXCHK(dmx_ctx_sync_checked(c));
    for (const Piece &pc : pieces)
        for (int r = 0; r < pc.rows; ++r)
            memcpy(out + pc.dst + (i64)r * pc.pitch, d.pinned + pc.stage + (i64)r * pc.len, sizeof(float) * (size_t)pc.len);
    return DMX_OK;
}

extern "C" int dmx_engine_track_infer(dmx_engine *e, const float *audio, int64_t n, const int *shift_offsets, float *out, int layout,
                                      dmx_progress_fn progress, void *user)
{
    if (!e || !audio || !out || n < 2)
        return dmx_fail(DMX_ERR_ARG, "dmx_engine_track_infer: invalid argument");
    if (layout != DMX_LAYOUT_EIGEN && layout != DMX_LAYOUT_PLANAR)
        return dmx_fail(DMX_ERR_ARG, "dmx_engine_track_infer: unknown layout %d", layout);
    std::lock_guard<std::mutex> guard(e->mu);
    const int G = (int)e->devs.size(), M = e->nModels, S = e->S;
    if (G == 1 && M == 1) // nothing to shard: the single-context path (same kernels, same bits; finishes the track in pieces)
        return dmx_track_infer(e->devs[0].ctx, audio, n, shift_offsets ? shift_offsets[0] : -1, out, layout, progress, user);
    const i64 seg = e->seg, blk = (i64)S * 2 * seg;
    std::vector<int> shifts((size_t)M), nseg((size_t)M);
    std::vector<i64> lens((size_t)M);
    i64 stride = 0;
    int T = 0;
    for (int m = 0; m < M; ++m)
    {
        int so = shift_offsets ? shift_offsets[m] : -1;
        if (so < 0)
            so = rand() % DMX_MAX_SHIFT; // one draw per model, in model order (demucs_ft.cpp:221-231 -> model_apply.cpp:114)
        if (so >= DMX_MAX_SHIFT)
            return dmx_fail(DMX_ERR_ARG, "dmx_engine_track_infer: shift_offset must be < %d", DMX_MAX_SHIFT);
        shifts[(size_t)m] = so;
        DMXCHK(dmx_track_geometry(e->devs[0].ctx, n, so, &lens[(size_t)m], &nseg[(size_t)m], &stride));
        T += nseg[(size_t)m];
    }
    std::vector<std::vector<Run>> runs;
    partition(nseg, G, runs);
    // the per-stem D2H of the OWNER mode needs each stem's samples contiguous: a bag in the Eigen image is finished on the root
    const bool owner = e->finish == DMX_FINISH_OWNER && G > 1 && (M == 1 || layout == DMX_LAYOUT_PLANAR);
    if (owner)
        for (int l = 0; l < G; ++l)
        {
            i64 slot = 0;
            for (size_t ri = 0; ri < runs[(size_t)l].size(); ++ri)
            {
                Run &r = runs[(size_t)l][ri];
                for (int q = l - 1; q >= 0 && r.g0 > 0 && r.predDev < 0; --q) // who owns segment g0-1 of this model
                    for (size_t rj = 0; rj < runs[(size_t)q].size(); ++rj)
                        if (runs[(size_t)q][rj].model == r.model && runs[(size_t)q][rj].g1 == r.g0)
                        {
                            r.predDev = q, r.predRun = (int)rj;
                            runs[(size_t)q][rj].succDev = l, runs[(size_t)q][rj].succRun = (int)ri;
                        }
                if (r.g0 > 0 && r.predDev < 0)
                    return dmx_fail(DMX_ERR_ARG, "dmx_engine_track_infer: internal error (segment %d of model %d has no owner)", r.g0 - 1, r.model);
                r.ownSlot = slot + (r.predDev >= 0 ? 1 : 0);
                slot = r.ownSlot + (r.g1 - r.g0);
            }
        }
    // ---- buffers (grown on first use, reused afterwards)
    for (int l = 0; l < G; ++l)
    {
        EngineDev &d = e->devs[(size_t)l];
        HIPCHK(hipSetDevice(d.dev));
        if (runs[(size_t)l].empty() && l != 0)
            continue;
        DMXCHK(dmx_ensure_buf(d.ctx->bAudio, 2 * n));
        if (layout == DMX_LAYOUT_PLANAR)
            DMXCHK(dmx_ensure_buf(d.ctx->bTmp, 2 * n));
        DMXCHK(dmx_ensure_buf(d.ctx->bMix, 2 * seg * e->maxBatch));
        if (owner)
        {
            const std::vector<Run> &rl = runs[(size_t)l];
            const i64 halo = (i64)S * 2 * (seg - stride);
            if (!rl.empty())
            {
                DMXCHK(dmx_ensure_buf(d.slab, (rl.back().ownSlot + (rl.back().g1 - rl.back().g0)) * blk));
                DMXCHK(dmx_ensure_buf(d.haloSend, (i64)rl.size() * halo));
                DMXCHK(dmx_ensure_buf(d.haloRecv, (i64)rl.size() * halo));
                DMXCHK(dmx_ensure_buf(d.ctx->bOut, (i64)S * 2 * n));
                // staging for the stretches this device finishes: <= (its share of a model's planes) x (its positions + slack)
                i64 need = 0;
                for (const Run &r : rl)
                    need += (i64)(M == 1 ? 2 * S : 2) * std::min<i64>(n, (i64)(r.g1 - r.g0) * stride + seg);
                if (need > d.pinnedCap)
                {
                    if (d.pinned)
                        HIPCHK(hipHostFree(d.pinned));
                    d.pinned = nullptr, d.pinnedCap = 0;
                    HIPCHK(hipHostMalloc((void **)&d.pinned, sizeof(float) * (size_t)need, hipHostMallocDefault));
                    d.pinnedCap = need;
                }
            }
        }
        else if (l != 0)
        {
            i64 items = 0;
            for (const Run &r : runs[(size_t)l])
                items += r.g1 - r.g0;
            DMXCHK(dmx_ensure_buf(d.slab, items * blk));
        }
    }
    HIPCHK(hipSetDevice(e->devs[0].dev));
    if (!owner)
    {
        for (int m = 0; m < M; ++m)
            DMXCHK(dmx_ensure_buf(e->segOut[(size_t)m], (i64)nseg[(size_t)m] * blk));
        DMXCHK(dmx_ensure_buf(e->out, (i64)S * 2 * n));
    }
    if (progress)
        progress(0.0f, "1., apply model w/ shift", user);
    // ---- one host thread per device; this thread delivers the progress lines (the reference invokes the
    // callback synchronously on the caller's thread, src/model.hpp:17)
    Shared sh;
    sh.workersLeft = G;
    sh.haloReady.resize((size_t)G);
    for (int l = 0; l < G; ++l)
        sh.haloReady[(size_t)l].assign(runs[(size_t)l].size(), 0);
    std::vector<std::thread> threads;
    for (int l = 0; l < G; ++l)
        threads.emplace_back([&, l] {
            EngineDev &d = e->devs[(size_t)l];
            d.rc = owner ? device_work_owner(e, l, audio, layout, n, shifts, nseg, lens, stride, runs, out, sh)
                         : device_work(e, l, audio, layout, n, shifts, nseg, runs[(size_t)l], runs, sh);
            d.err = d.rc == DMX_OK ? std::string() : dmx_err_string();
            std::lock_guard<std::mutex> lk(sh.mu);
            --sh.workersLeft;
            if (d.rc != DMX_OK)
                sh.failed = true;
            sh.cv.notify_all();
        });
    {
        std::unique_lock<std::mutex> lk(sh.mu);
        for (;;)
        {
            sh.cv.wait(lk, [&] { return !sh.messages.empty() || sh.workersLeft == 0; });
            std::vector<std::string> msgs;
            msgs.swap(sh.messages);
            const float frac = T > 0 ? (float)sh.itemsDone / (float)T : 1.0f;
            const bool finished = sh.workersLeft == 0;
            lk.unlock();
            if (progress)
                for (const std::string &s : msgs)
                    progress(frac, s.c_str(), user);
            lk.lock();
            if (finished && sh.messages.empty())
                break;
        }
    }
    for (std::thread &t : threads)
        t.join();
    for (int l = 0; l < G; ++l)
        if (e->devs[(size_t)l].rc != DMX_OK)
        {
            dmx_set_err_string("device " + std::to_string(l) + ": " + e->devs[(size_t)l].err);
            return e->devs[(size_t)l].rc;
        }
    if (owner) // every device has finished and copied out its own stretch
        return DMX_OK;
    // ---- root: overlap-add per model in segment order, stem m from model m for a bag
    // (model_apply.cpp:171-246; demucs_ft.cpp:238-241), de-normalise, copy out
    EngineDev &r0 = e->devs[0];
    dmx_ctx *c = r0.ctx;
    HIPCHK(hipSetDevice(r0.dev));
    for (int m = 0; m < M; ++m)
    {
        DMXCHK(dmx_ctx_set_model(c, r0.models[(size_t)m]));
        DMXCHK(dmx_track_overlap_add_planes(c, e->segOut[(size_t)m].p, nseg[(size_t)m], n, shifts[(size_t)m], c->dStats, e->out.p, layout,
                                            M == 1 ? 0 : 2 * m, M == 1 ? 2 * S : 2));
    }
    HIPCHK(hipMemcpyAsync(out, e->out.p, sizeof(float) * (size_t)S * 2 * n, hipMemcpyDeviceToHost, c->stream));
    DMXCHK(dmx_ctx_sync_checked(c));
    return DMX_OK;
}
