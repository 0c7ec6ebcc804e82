#include "knobs.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace ptr {

namespace {

long long numberOr(const char* name, long long fallback) {
    const char* e = std::getenv(name);
    if (!e || !*e) return fallback;
    char* end = nullptr;
    const long long v = std::strtoll(e, &end, 10);
    return end == e ? fallback : v;
}

bool hasTopic(const char* list, const char* topic) {
    if (!list) return false;
    const size_t n = std::strlen(topic);
    for (const char* p = list; *p;) {
        const char* q = std::strchr(p, ',');
        const size_t len = q ? static_cast<size_t>(q - p) : std::strlen(p);
        if ((len == n && std::strncmp(p, topic, n) == 0) || (len == 3 && std::strncmp(p, "all", 3) == 0)) return true;
        p += len + (q ? 1 : 0);
    }
    return false;
}

}  // namespace

Knobs readKnobs() {
    Knobs k;
    k.poolSlots = static_cast<uint64_t>(std::min<long long>(std::max<long long>(numberOr("PTR_POOL_SLOTS", 0), 0), 64ll << 20));
    if (k.poolSlots != 0 && k.poolSlots < 1024) k.poolSlots = 1024;
    k.poolGroups = static_cast<uint32_t>(std::min<long long>(std::max<long long>(numberOr("PTR_POOL_GROUPS", 0), 0), 8));
    k.connectOverlap = static_cast<int>(numberOr("PTR_CONNECT_OVERLAP", -1));
    k.wideNodes = static_cast<int>(numberOr("PTR_WIDE_NODES", -1));
    k.quantizedNodes = static_cast<int>(numberOr("PTR_QUANTIZED_NODES", -1));
    k.tailBelow = numberOr("PTR_TAIL_BELOW", -1);
    k.maxItems = static_cast<uint64_t>(std::max<long long>(numberOr("PTR_MAX_ITEMS", 0), 0));
    if (k.maxItems != 0 && k.maxItems < 1024) k.maxItems = 1024;
    k.refillBelow = static_cast<int>(std::min<long long>(std::max<long long>(numberOr("PTR_REFILL_BELOW", 0), 0), 64));
    k.buildThreads = static_cast<uint32_t>(std::min<long long>(std::max<long long>(numberOr("PTR_BUILD_THREADS", 0), 0), 256));
    k.noOversize = numberOr("PTR_NO_OVERSIZE", 0) != 0;
    const char* verbose = std::getenv("PTR_VERBOSE");
    k.verboseBuild = hasTopic(verbose, "build");
    k.verbosePolls = hasTopic(verbose, "polls");
    k.verboseLaunches = hasTopic(verbose, "launches");
    k.verboseSteps = hasTopic(verbose, "steps");
    return k;
}

}  // namespace ptr
