// Every environment variable the library reads, in one place.  They exist for tests (scheduling must not change the image:
// tests/test_gpu_parity.py::test_scheduling_knobs_do_not_change_the_image sets each of them) and for A/B measurements; a product run
// sets none.  Read afresh at every scene upload / render, so a test can change them between two calls of one process.
#pragma once

#include <cstdint>

namespace ptr {

struct Knobs {
    uint64_t poolSlots = 0;        // PTR_POOL_SLOTS        resident path slots at most (0: default 32 Mi; capped at 64 Mi)
    uint32_t poolGroups = 0;       // PTR_POOL_GROUPS       concurrent groups of the pool, 1..8 (0: default - 2, or 4 for pools of at most 8 Mi slots)
    int connectOverlap = -1;       // PTR_CONNECT_OVERLAP   0: k_connect on its group's own stream (-1: default, beside the next k_extend when the pool runs as <= 2 groups)
    int wideNodes = -1;            // PTR_WIDE_NODES        0: the persistent kernels walk the binary nodes; 2: four-wide nodes collapsed by level (-1: default, four-wide by area)
    int quantizedNodes = -1;       // PTR_QUANTIZED_NODES   0 / 1: force 64 B float / 32 B quantised nodes (-1: decided by the scene's grid)
    int64_t tailBelow = -1;        // PTR_TAIL_BELOW        live slots below which the end-of-frame kernels take over (0: never; -1: default)
    uint64_t maxItems = 0;         // PTR_MAX_ITEMS         per-sample accumulators one pass may hold (0: from the device's memory)
    int refillBelow = 0;           // PTR_REFILL_BELOW      traversing lanes below which a persistent wave refills, 1..64 (0: default 40)
    uint32_t buildThreads = 0;     // PTR_BUILD_THREADS     BVH builder threads (0: all cores)
    bool noOversize = false;       // PTR_NO_OVERSIZE       keep every triangle in the tree
    // PTR_VERBOSE: comma-separated topics printed to stderr - build (BVH / upload timings), polls (live slots per host poll),
    // launches (when each kernel ran), steps (lane-utilisation counters of a counting render)
    bool verboseBuild = false, verbosePolls = false, verboseLaunches = false, verboseSteps = false;
};

Knobs readKnobs();

}  // namespace ptr
