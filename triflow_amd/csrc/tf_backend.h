// Device back end used by the host runtime (tf_solver.h): memory, stream, code-object loading,
// kernel launch, event timing.  tf_backend_hip.cpp is the product
// implementation (HIP on gfx950).  tests/emu/tf_backend_emu.cpp is a test-only
// stand-in that executes the same kernel bodies thread by thread on the host so
// that the host orchestration can be exercised without a GPU; it is never part
// of libtriflow_hip.so.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string>

namespace tfb {

struct Module;      // loaded code object + kernel handles
struct Stream;
struct Event;

bool is_device_build();
// lanes that share one chunk in the reduced-level kernels of block size b
int coop_group(int b);
// reduced levels of block size b run the cyclic-reduction kernels (tfk_cr_*)
bool cyclic_reduction(int b);
int device_count();
void set_device(int ordinal);                 // throws std::runtime_error

void* dev_alloc(size_t bytes);                // zero-initialised
void dev_free(void* p);
void memset0(void* p, size_t bytes, Stream* s);
void h2d(void* dst, const void* src, size_t bytes, Stream* s);     // synchronous w.r.t. host
void d2h(void* dst, const void* src, size_t bytes, Stream* s);     // synchronous w.r.t. host
void d2d(void* dst, const void* src, size_t bytes, Stream* s);

Module* module_load(const void* image, size_t bytes);
// kernels whose bit is set in `mask` (kernel table index) are taken from a second build of the
// same translation unit (the per-kernel optimisation gate of compilers.py)
void module_add_alternate(Module* m, const void* image, size_t bytes, uint64_t mask);
void module_unload(Module* m);

Stream* stream_create();
void stream_destroy(Stream* s);
void stream_sync(Stream* s);

// the workgroup size a kernel of the code object was compiled for (its launch bound): some
// kernels exist in a one- and a two-wavefront form per model (tfk_l1_factor*)
unsigned kernel_block(Module* m, int kernel);

// one launch: grid (gx, gy, 1), block (block, 1, 1), one by-value argument struct;
// lds_bytes: dynamic LDS of the workgroup (extern __shared__)
void launch(Module* m, int kernel, unsigned gx, unsigned gy, unsigned block,
            const void* args, size_t arg_bytes, Stream* s, unsigned lds_bytes = 0);
// same launch; `start` / `stop` receive the kernel's own begin / end timestamps
// (not the stream-order interval around it)
void launch_timed(Module* m, int kernel, unsigned gx, unsigned gy, unsigned block,
                  const void* args, size_t arg_bytes, Stream* s, Event* start, Event* stop,
                  unsigned lds_bytes = 0);

// Stream capture into an executable graph (HIP graphs): a step of a small problem is a string of
// launch-bound kernels, replayed with one call.  The emulation has none (graphs_supported()).
struct Graph;
bool graphs_supported();
void capture_begin(Stream* s);
Graph* capture_end(Stream* s);               // ends the capture, instantiates
void capture_abort(Stream* s);
void graph_launch(Graph* g, Stream* s);
void graph_destroy(Graph* g);

// A few bytes from the device to the host without waiting for what is queued behind them: `post` queues
// the copy (page-locked target) and marks the spot in the stream, `wait` blocks until that spot is
// reached -- not until the stream is empty, as d2h does -- and hands the bytes over (<= 64).
struct Mailbox;
Mailbox* mailbox_create();
void mailbox_destroy(Mailbox* m);
void mailbox_post(Mailbox* m, int at, const void* dev_src, size_t bytes, Stream* s);   // bytes land at offset `at`
void mailbox_mark(Mailbox* m, Stream* s);
void mailbox_wait(Mailbox* m, void* dst, size_t bytes);

Event* event_create();
void event_destroy(Event* e);
void event_record(Event* e, Stream* s);
float event_elapsed_ms(Event* a, Event* b);   // both completed

}  // namespace tfb
