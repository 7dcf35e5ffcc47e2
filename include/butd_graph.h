/*
 * butd_graph.h -- C ABI of the hipGraph hygiene pass of the training step (gfx950, ROCm 7.2).
 *
 * The reference's step is eager PyTorch (main_utils.py:415-438); this build replays it as hipGraphs
 * (butd_detr_amd/train_step.py).  On ROCm 7.2 a MEMSET node of a replayed graph is unreliable: replayed behind a
 * still-running graph (any size) or a second time (multi-megabyte ranges) it writes a garbage pattern instead of
 * its value (scratch/graph_node_order.py) -- which silently breaks every captured
 * hipMemsetAsync, e.g. the semaphore reset of torch's multi-block reductions.  Kernel and memcpy nodes are sound.
 */
#ifndef BUTD_GRAPH_H
#define BUTD_GRAPH_H
#ifdef __cplusplus
extern "C" {
#endif

/* Rewrites every MEMSET node of `graph` (a hipGraph_t that has not been instantiated yet) into a KERNEL node with
 * the same destination, value, element size (1, 2 or 4 bytes), width, height and pitch, the same dependencies and
 * the same dependents, and destroys the memset node.  *replaced receives the number of rewritten nodes.
 * Returns 0 or a hipError_t. */
int butd_graph_replace_memset_nodes(void *graph, int *replaced);

/* Number of nodes of `graph` by hipGraphNodeType (counts[0..15], others ignored); the nodes of embedded child
 * graphs (hipGraphNodeTypeGraph) are counted too, and butd_graph_replace_memset_nodes rewrites inside them as well. */
int butd_graph_node_counts(void *graph, int counts[16]);

/* Profiler-free timeline: a one-wave kernel that writes the device's constant-rate wall clock (s_memrealtime, 100 MHz)
 * into slots[slot] when the stream reaches it.  Captured into the step's hipGraph at region boundaries it gives the
 * in-situ duration of each region without rocprofv3 (which serialises the queues of a multi-stream graph);
 * scratch/step_marks.py.  Not on the product path. */
int butd_timeline_mark(unsigned long long *slots, int slot, void *stream);

/* hipRuntimeGetVersion / hipDriverGetVersion of the process: recorded next to the memset-node finding (the bug was
 * seen on HIP runtime 7.2, torch 2.10.0+rocm7.0; tests/test_gpu_runtime_probe.py reports whether it is still there). */
int butd_runtime_versions(int *runtime, int *driver);

/* A HIP stream that belongs to the caller alone (hipStreamCreateWithPriority, non-blocking; priority 0 = default,
 * negative = higher).  torch.cuda.Stream() hands out the members of a fixed pool of 32 streams per priority round-robin,
 * the pool RCCL's own stream (ProcessGroupNCCL) and torch's default graph-capture stream come from too: the 33rd
 * "new" stream of a process IS one of the first 32 again.  A forked branch of a captured step that lands on RCCL's stream
 * puts that stream into capture mode while the process group's watchdog thread polls events recorded on it -> the
 * process aborts with hipErrorCapturedEvent (round 4, DESIGN.md 7.4 #6b; reproducer: scratch/stream_alias_probe.py).
 * The step's streams therefore never come from the pool (butd_detr_amd/graph_audit.py: own_stream). */
int butd_stream_create(int priority, void **stream);
int butd_stream_destroy(void *stream);

#ifdef __cplusplus
}
#endif
#endif
