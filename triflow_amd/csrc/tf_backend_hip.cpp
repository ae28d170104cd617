// HIP implementation of tf_backend.h (gfx950 / MI355X).
#include "tf_backend.h"
#include "tf_args.h"

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>

namespace tfb {

static void check(hipError_t err, const char* what) {
    if (err != hipSuccess)
        throw std::runtime_error(std::string("HIP error in ") + what + ": " + hipGetErrorString(err));
}
#define TF_HIP(call) check((call), #call)

struct Module { hipModule_t mod; hipFunction_t fn[TFK_COUNT]; };
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
void h2d(void* dst, const void* src, size_t bytes, Stream* s) {
    TF_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s->s));
    TF_HIP(hipStreamSynchronize(s->s));
}
void d2h(void* dst, const void* src, size_t bytes, Stream* s) {
    TF_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, s->s));
    TF_HIP(hipStreamSynchronize(s->s));
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
void module_unload(Module* m) {
    if (!m) return;
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

void launch(Module* m, int kernel, unsigned gx, unsigned gy, unsigned block,
            const void* args, size_t arg_bytes, Stream* s) {
    size_t size = arg_bytes;
    void* config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, const_cast<void*>(args),
                      HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
    TF_HIP(hipModuleLaunchKernel(m->fn[kernel], gx, gy, 1, block, 1, 1, 0, s->s, nullptr, config));
}

void launch_timed(Module* m, int kernel, unsigned gx, unsigned gy, unsigned block,
                  const void* args, size_t arg_bytes, Stream* s, Event* start, Event* stop) {
    size_t size = arg_bytes;
    void* config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, const_cast<void*>(args),
                      HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
    // hipExtModuleLaunchKernel takes the global size in work-items
    TF_HIP(hipExtModuleLaunchKernel(m->fn[kernel], gx * block, gy, 1, block, 1, 1, 0, s->s,
                                    nullptr, config, start->e, stop->e, 0));
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
