// shim/shim_error.h -- ONE error convention for every member function the shim replaces.
//
// The reference's ORBextractor / ORBmatcher / Frame / KeyFrame::ComputeBoW / Optimizer bodies have no error channel and never throw
// ("void/int returns, no exceptions"); their callers run on the tracking and the local-mapping threads (src/Tracking.cc,
// src/LocalMapping.cc:123), where an escaping exception is std::terminate.  A device error inside a replaced body therefore
//   - is written to std::cerr (as the reference reports its own failures, src/System.cc:61),
//   - is counted and kept: orbx_shim_error_count() / orbx_shim_last_error() (process wide, all five shim files),
//   - and the body returns the reference's "nothing found" value: 0 matches / outputs as the reference leaves them for an empty result /
//     early return;
//   - ORBX_SHIM_FATAL=1 (read once) turns it into a std::runtime_error instead (tests, bring-up).
// Header only (function-local statics have one instance per linked image), so a build that swaps a single file of the shim needs nothing else.
#ifndef ORBX_SHIM_ERROR_H
#define ORBX_SHIM_ERROR_H

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <mutex>
#include <stdexcept>
#include <string>

#include "orbx.h"

namespace orbx_shim
{
struct ErrorState {
    std::atomic<long> count;
    std::mutex m;
    std::string last;
    ErrorState() : count(0) {}
};
inline ErrorState &Errors() { static ErrorState s; return s; }
// device of every liborbx handle the shim creates (ORBextractor::SetDevice; default 0): the matcher / frame / vocabulary / optimizer handles read the
// extractor's device buffers in place, so they must live on its device
inline int &Device() { static int d = 0; return d; }
inline bool Fatal() { static const bool f = [] { const char *e = getenv("ORBX_SHIM_FATAL"); return e && e[0] == '1'; }(); return f; }

// records `where (orbx): <library message>` (or `detail` when the failure is the shim's own), returns false
inline bool Fail(const char *where, const char *detail = 0)
{
    const std::string msg = std::string(where) + " (orbx): " + (detail ? detail : orbx_last_error());
    ErrorState &e = Errors();
    {
        std::lock_guard<std::mutex> lock(e.m);
        e.last = msg;
    }
    e.count.fetch_add(1);
    std::cerr << msg << std::endl;
    if (Fatal()) throw std::runtime_error(msg);
    return false;
}
}  // namespace orbx_shim

// Timeline marks (shim/Frame_hip.cc defines the recorder; a build without that file has none): orbx_shim::Mark("name") costs one relaxed load when off.
extern "C" __attribute__((weak)) void orbx_shim_trace_mark(const char *name);
namespace orbx_shim { inline void Mark(const char *name) { if (orbx_shim_trace_mark) orbx_shim_trace_mark(name); } }

extern "C" {
__attribute__((weak, visibility("default"))) long orbx_shim_error_count(void) { return orbx_shim::Errors().count.load(); }
// the last message, copied into the caller's buffer (always terminated); returns its full length
__attribute__((weak, visibility("default"))) int orbx_shim_last_error(char *buf, int capacity)
{
    orbx_shim::ErrorState &e = orbx_shim::Errors();
    std::lock_guard<std::mutex> lock(e.m);
    if (buf && capacity > 0) { strncpy(buf, e.last.c_str(), (size_t)capacity - 1); buf[capacity - 1] = 0; }
    return (int)e.last.size();
}
}

#endif
