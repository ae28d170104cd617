// HIP implementation of tf_backend.h (gfx950 / MI355X).
#include "tf_backend.h"
#include "tf_args.h"

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <thread>
#include <stdexcept>
#include <string>

namespace tfb {

static void check(hipError_t err, const char* what) {
    if (err != hipSuccess)
        throw std::runtime_error(std::string("HIP error in ") + what + ": " + hipGetErrorString(err));
}
#define TF_HIP(call) check((call), #call)

struct Module { hipModule_t mod; hipFunction_t fn[TFK_COUNT]; hipModule_t alt = nullptr; };
struct Stream { hipStream_t s; };
struct Event { hipEvent_t e; };

bool is_device_build() { return true; }
int coop_group(int b) { return b <= 2 ? 1 : (b <= 8 ? 8 : (b <= 16 ? 16 : 1)); }   // TfCoop<b>::G

bool cyclic_reduction(int b) {
    // TRIFLOW_REDUCED=walk selects the chunk walks of tfk_bt_* instead (A/B comparisons)
    const char* mode = getenv("TRIFLOW_REDUCED");
    if (mode && std::string(mode) == "walk") return false;
    return b >= 1 && b <= 8;
}

int device_count() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

void set_device(int ordinal) { TF_HIP(hipSetDevice(ordinal)); }

void* dev_alloc(size_t bytes) {
    void* p = nullptr;
    if (bytes == 0) bytes = 8;
    TF_HIP(hipMalloc(&p, bytes));
    // the solver streams are non-blocking: make sure the zero fill (null stream)
    // has landed before anything is uploaded into the new buffer
    TF_HIP(hipMemset(p, 0, bytes));
    TF_HIP(hipDeviceSynchronize());
    return p;
}
void dev_free(void* p) { if (p) (void)hipFree(p); }
void memset0(void* p, size_t bytes, Stream* s) { TF_HIP(hipMemsetAsync(p, 0, bytes, s->s)); }
// Host <-> device copies of the drop-in path (model.F / model.J: NumPy arrays in and out).  A
// copy from / to pageable memory is staged by the runtime in small pieces; large transfers go
// through two page-locked bounce buffers of our own instead: the CPU fills (or drains) one
// with several threads while the DMA engine moves the other.
namespace {
constexpr size_t kStageChunk = 8u << 20;        // bytes per bounce buffer
constexpr size_t kStageMin = 4u << 20;          // smaller transfers: plain hipMemcpyAsync
struct Bounce {
    void* buf[2] = {nullptr, nullptr};
    hipEvent_t ev[2];
    bool ok = false;
    Bounce() {
        if (hipHostMalloc(&buf[0], kStageChunk, hipHostMallocDefault) != hipSuccess) return;
        if (hipHostMalloc(&buf[1], kStageChunk, hipHostMallocDefault) != hipSuccess) return;
        if (hipEventCreateWithFlags(&ev[0], hipEventDisableTiming) != hipSuccess) return;
        if (hipEventCreateWithFlags(&ev[1], hipEventDisableTiming) != hipSuccess) return;
        ok = true;
    }
};
// one set per (thread, device): the events belong to the device that was current when they were
// made, and a thread may drive solvers on several GPUs (SolverOpts.device); released with the thread
struct BounceSet {
    std::map<int, Bounce*> by_device;
    ~BounceSet() {
        for (auto& kv : by_device) {
            Bounce* b = kv.second;
            if (b->ok) { (void)hipEventDestroy(b->ev[0]); (void)hipEventDestroy(b->ev[1]); }
            if (b->buf[0]) (void)hipHostFree(b->buf[0]);
            if (b->buf[1]) (void)hipHostFree(b->buf[1]);
            delete b;
        }
    }
};
Bounce& bounce() {
    static thread_local BounceSet set;
    int dev = 0;
    (void)hipGetDevice(&dev);
    Bounce*& b = set.by_device[dev];
    if (!b) b = new Bounce();
    return *b;
}
void parallel_copy(void* dst, const void* src, size_t n) {
    constexpr int NT = 4;
    if (n < (1u << 20)) { std::memcpy(dst, src, n); return; }
    std::thread th[NT - 1];
    const size_t part = (n / NT + 63) & ~(size_t)63;
    for (int t = 1; t < NT; ++t) {
        const size_t lo = std::min(n, part * t), hi = std::min(n, part * (t + 1));
        th[t - 1] = std::thread([=] { if (hi > lo) std::memcpy((char*)dst + lo, (const char*)src + lo, hi - lo); });
    }
    std::memcpy(dst, src, std::min(n, part));
    for (int t = 1; t < NT; ++t) th[t - 1].join();
}
}  // namespace
void h2d(void* dst, const void* src, size_t bytes, Stream* s) {
    Bounce& b = bounce();
    if (bytes < kStageMin || !b.ok) {
        TF_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s->s));
        TF_HIP(hipStreamSynchronize(s->s));
        return;
    }
    size_t off = 0;
    for (int k = 0; off < bytes; ++k) {
        const int i = k & 1;
        const size_t n = std::min(kStageChunk, bytes - off);
        if (k >= 2) TF_HIP(hipEventSynchronize(b.ev[i]));           // the DMA out of this buffer is done
        parallel_copy(b.buf[i], (const char*)src + off, n);
        TF_HIP(hipMemcpyAsync((char*)dst + off, b.buf[i], n, hipMemcpyHostToDevice, s->s));
        TF_HIP(hipEventRecord(b.ev[i], s->s));
        off += n;
    }
    TF_HIP(hipStreamSynchronize(s->s));
}
void d2h(void* dst, const void* src, size_t bytes, Stream* s) {
    Bounce& b = bounce();
    if (bytes < kStageMin || !b.ok) {
        TF_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, s->s));
        TF_HIP(hipStreamSynchronize(s->s));
        return;
    }
    size_t off = 0, done = 0;
    int k = 0;
    for (; off < bytes; ++k) {
        const int i = k & 1;
        const size_t n = std::min(kStageChunk, bytes - off);
        if (k >= 2) {                                              // drain the chunk that used this buffer
            TF_HIP(hipEventSynchronize(b.ev[i]));
            const size_t m = std::min(kStageChunk, bytes - done);
            parallel_copy((char*)dst + done, b.buf[i], m);
            done += m;
        }
        TF_HIP(hipMemcpyAsync(b.buf[i], (const char*)src + off, n, hipMemcpyDeviceToHost, s->s));
        TF_HIP(hipEventRecord(b.ev[i], s->s));
        off += n;
    }
    for (int j = (k >= 2 ? k - 2 : 0); j < k; ++j) {               // the last one or two chunks
        const int i = j & 1;
        TF_HIP(hipEventSynchronize(b.ev[i]));
        const size_t m = std::min(kStageChunk, bytes - done);
        parallel_copy((char*)dst + done, b.buf[i], m);
        done += m;
    }
}
void d2d(void* dst, const void* src, size_t bytes, Stream* s) {
    TF_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s->s));
}

Module* module_load(const void* image, size_t) {
    static const char* names[TFK_COUNT] = TF_KERNEL_NAMES;
    Module* m = new Module();
    hipError_t err = hipModuleLoadData(&m->mod, image);
    if (err != hipSuccess) {
        delete m;
        check(err, "hipModuleLoadData (is the code object built for this GPU, gfx950?)");
    }
    for (int k = 0; k < TFK_COUNT; ++k) {
        err = hipModuleGetFunction(&m->fn[k], m->mod, names[k]);
        if (err != hipSuccess) {
            (void)hipModuleUnload(m->mod);
            delete m;
            throw std::runtime_error(std::string("kernel missing from code object: ") + names[k]);
        }
    }
    // Debug aid for compiler bisection: TF_ALT_HSACO=<file> TF_ALT_MASK=<bits> takes
    // the kernels whose bit is set from a second build of the same model.
    if (const char* alt = getenv("TF_ALT_HSACO")) {
        const char* mask_s = getenv("TF_ALT_MASK");
        unsigned long mask = mask_s ? strtoul(mask_s, nullptr, 0) : 0;
        FILE* f = fopen(alt, "rb");
        if (f && mask) {
            fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
            std::string img(n, '\0');
            if (fread(&img[0], 1, n, f) == (size_t)n) {
                hipModule_t am;
                if (hipModuleLoadData(&am, img.data()) == hipSuccess)
                    for (int k = 0; k < TFK_COUNT; ++k)
                        if ((mask >> k) & 1ul) (void)hipModuleGetFunction(&m->fn[k], am, names[k]);
            }
        }
        if (f) fclose(f);
    }
    return m;
}
void module_add_alternate(Module* m, const void* image, size_t, uint64_t mask) {
    static const char* names[TFK_COUNT] = TF_KERNEL_NAMES;
    if (!m || !mask) return;
    if (m->alt) throw std::runtime_error("module_add_alternate: one alternate build per model");
    check(hipModuleLoadData(&m->alt, image), "hipModuleLoadData (alternate build)");
    for (int k = 0; k < TFK_COUNT; ++k)
        if ((mask >> k) & 1ull) {
            if (hipModuleGetFunction(&m->fn[k], m->alt, names[k]) != hipSuccess)
                throw std::runtime_error(std::string("kernel missing from the alternate code object: ") + names[k]);
        }
}
void module_unload(Module* m) {
    if (!m) return;
    if (m->alt) (void)hipModuleUnload(m->alt);
    (void)hipModuleUnload(m->mod);
    delete m;
}

Stream* stream_create() {
    Stream* s = new Stream();
    TF_HIP(hipStreamCreateWithFlags(&s->s, hipStreamNonBlocking));
    return s;
}
void stream_destroy(Stream* s) { if (s) { (void)hipStreamDestroy(s->s); delete s; } }
void stream_sync(Stream* s) { TF_HIP(hipStreamSynchronize(s->s)); }

unsigned kernel_block(Module* m, int kernel) {
    int v = 0;
    TF_HIP(hipFuncGetAttribute(&v, HIP_FUNC_ATTRIBUTE_MAX_THREADS_PER_BLOCK, m->fn[kernel]));
    return (unsigned)v;
}

void launch(Module* m, int kernel, unsigned gx, unsigned gy, unsigned block,
            const void* args, size_t arg_bytes, Stream* s, unsigned lds_bytes) {
    size_t size = arg_bytes;
    void* config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, const_cast<void*>(args),
                      HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
    TF_HIP(hipModuleLaunchKernel(m->fn[kernel], gx, gy, 1, block, 1, 1, lds_bytes, s->s, nullptr, config));
}

void launch_timed(Module* m, int kernel, unsigned gx, unsigned gy, unsigned block,
                  const void* args, size_t arg_bytes, Stream* s, Event* start, Event* stop,
                  unsigned lds_bytes) {
    size_t size = arg_bytes;
    void* config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, const_cast<void*>(args),
                      HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
    // hipExtModuleLaunchKernel takes the global size in work-items
    TF_HIP(hipExtModuleLaunchKernel(m->fn[kernel], gx * block, gy, 1, block, 1, 1, lds_bytes, s->s,
                                    nullptr, config, start->e, stop->e, 0));
}

struct Graph { hipGraph_t graph; hipGraphExec_t exec; };
bool graphs_supported() { return true; }
void capture_begin(Stream* s) { TF_HIP(hipStreamBeginCapture(s->s, hipStreamCaptureModeThreadLocal)); }
Graph* capture_end(Stream* s) {
    hipGraph_t graph = nullptr;
    TF_HIP(hipStreamEndCapture(s->s, &graph));
    hipGraphExec_t exec = nullptr;
    hipError_t err = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    if (err != hipSuccess) { (void)hipGraphDestroy(graph); check(err, "hipGraphInstantiate"); }
    return new Graph{graph, exec};
}
void capture_abort(Stream* s) {
    hipGraph_t graph = nullptr;
    (void)hipStreamEndCapture(s->s, &graph);
    if (graph) (void)hipGraphDestroy(graph);
}
void graph_launch(Graph* g, Stream* s) { TF_HIP(hipGraphLaunch(g->exec, s->s)); }
void graph_destroy(Graph* g) {
    if (!g) return;
    (void)hipGraphExecDestroy(g->exec);
    (void)hipGraphDestroy(g->graph);
    delete g;
}

struct Mailbox { char* host = nullptr; hipEvent_t ev; };
Mailbox* mailbox_create() {
    Mailbox* m = new Mailbox();
    TF_HIP(hipHostMalloc((void**)&m->host, 64, hipHostMallocDefault));
    std::memset(m->host, 0, 64);
    TF_HIP(hipEventCreateWithFlags(&m->ev, hipEventDisableTiming));
    return m;
}
void mailbox_destroy(Mailbox* m) {
    if (!m) return;
    (void)hipEventDestroy(m->ev);
    (void)hipHostFree(m->host);
    delete m;
}
void mailbox_post(Mailbox* m, int at, const void* dev_src, size_t bytes, Stream* s) {
    TF_HIP(hipMemcpyAsync(m->host + at, dev_src, bytes, hipMemcpyDeviceToHost, s->s));
}
void mailbox_mark(Mailbox* m, Stream* s) { TF_HIP(hipEventRecord(m->ev, s->s)); }
void mailbox_wait(Mailbox* m, void* dst, size_t bytes) {
    TF_HIP(hipEventSynchronize(m->ev));
    std::memcpy(dst, m->host, bytes);
}

Event* event_create() {
    Event* e = new Event();
    TF_HIP(hipEventCreate(&e->e));
    return e;
}
void event_destroy(Event* e) { if (e) { (void)hipEventDestroy(e->e); delete e; } }
void event_record(Event* e, Stream* s) { TF_HIP(hipEventRecord(e->e, s->s)); }
float event_elapsed_ms(Event* a, Event* b) {
    float ms = 0.f;
    TF_HIP(hipEventElapsedTime(&ms, a->e, b->e));
    return ms;
}

}  // namespace tfb
