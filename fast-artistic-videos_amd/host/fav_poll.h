// fav_poll.h -- waiting for files a producer is still writing (shared by fav_stylize and fav_stylize_vr).
//
// The reference's consumer (fast_artistic_video/utils.lua:74-80): if the file is not there, poll once a second until it is, then
// `sleep 1` once more.  That extra second is what makes its own pipeline work: stylizeVideo_deepflow.sh:83-96 runs
// makeOptFlow_deepflow.sh in the background next to the stylizer, and the checker that script calls writes a FULL-SIZE ALL-255
// PLACEHOLDER first (consistencyChecker/consistencyChecker.cpp:151-152) and the real mask over it ~0.1 s later (:171, fopen "wb":
// the file is empty, then short, then complete).  A consumer that takes a file as soon as it exists with a stable size therefore
// stylises against "everything reliable".
//
// Rule here (-poll_settle <s>, default 1.0 = the reference's second):
//   * a file whose last modification is at least <settle> seconds old is taken at once (finished inputs cost nothing);
//   * anything younger -- it appeared while we waited, or it was there but fresh -- is taken only once it has not been modified
//     for <settle> seconds (by its mtime against the wall clock, or, for file systems whose clock is not ours, by our own clock:
//     size and mtime unchanged for <settle> seconds of observation);
//   * a .pgm / .ppm / .flo that then still reads short (payload shorter than its header promises, or no header yet) is polled again
//     instead of being an error for as long as its producer may still be at work (modified less than max(5, 5 x settle) s ago);
//   * everything is bounded by -poll_timeout.
#pragma once
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <string>

#include "../../include/fav.h"

namespace favp {

struct Poll {
    double timeout_s = 600.0;   // -poll_timeout
    double settle_s = 1.0;      // -poll_settle
};

enum WaitResult { WAIT_OK = 0, WAIT_TIMEOUT = 1 };

inline double wall_age_seconds(const struct stat& st)
{
    struct timespec now; clock_gettime(CLOCK_REALTIME, &now);
    return (double)(now.tv_sec - st.st_mtim.tv_sec) + 1e-9 * (double)(now.tv_nsec - st.st_mtim.tv_nsec);
}

// Returns WAIT_OK once `path` exists, is not empty and has settled (see above); WAIT_TIMEOUT after timeout_s.
// Prints the reference's `Waiting for file "<path>"` line (utils.lua:76) once when the file is not there at the first look.
inline WaitResult wait_for_file(const std::string& path, const Poll& p)
{
    using clk = std::chrono::steady_clock;
    const auto t0 = clk::now();
    bool announced = false;
    // what we have observed ourselves: (size, mtime) and since when it has not changed
    long long seen_size = -1; struct timespec seen_mtime = {0, 0}; clk::time_point seen_since = t0;
    for (;;) {
        struct stat st;
        double remaining = 0.05;
        if (stat(path.c_str(), &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) {
            const double age = wall_age_seconds(st);
            if (age >= p.settle_s) return WAIT_OK;
            const auto now = clk::now();
            if ((long long)st.st_size != seen_size || st.st_mtim.tv_sec != seen_mtime.tv_sec || st.st_mtim.tv_nsec != seen_mtime.tv_nsec) {
                seen_size = (long long)st.st_size; seen_mtime = st.st_mtim; seen_since = now;
            }
            const double watched = std::chrono::duration<double>(now - seen_since).count();
            if (watched >= p.settle_s) return WAIT_OK;                       // a clock that is not ours (mtime in the future)
            remaining = p.settle_s - std::max(age, watched);                 // until the earlier of the two acceptances
        } else {
            seen_size = -1;
            if (!announced) { printf("Waiting for file \"%s\"\n", path.c_str()); fflush(stdout); announced = true; }
        }
        if (std::chrono::duration<double>(clk::now() - t0).count() > p.timeout_s) return WAIT_TIMEOUT;
        const double nap = std::min(0.05, std::max(0.001, remaining + 0.0005));
        usleep((useconds_t)(nap * 1e6));
    }
}

// Waits for `path`, then runs `reader()` (one of the fav_read_*_host calls; returns a fav_status).  FAV_EFORMAT -- a payload shorter
// than the header promises, or a header that is not there yet -- polls again while the producer may still be writing.
// Returns the reader's status, or FAV_EIO with `*timed_out = true` when the file never settled within the time-out.
template <class Reader>
inline int read_when_complete(const std::string& path, const Poll& p, Reader&& reader, bool* timed_out)
{
    using clk = std::chrono::steady_clock;
    const auto t0 = clk::now();
    const double grace = std::max(5.0, 5.0 * p.settle_s);
    *timed_out = false;
    for (;;) {
        Poll left = p;
        left.timeout_s = p.timeout_s - std::chrono::duration<double>(clk::now() - t0).count();
        if (left.timeout_s <= 0 || wait_for_file(path, left) != WAIT_OK) { *timed_out = true; return FAV_EIO; }
        const int rc = reader();
        if (rc != FAV_EFORMAT) return rc;
        struct stat st;
        if (stat(path.c_str(), &st) != 0) continue;                          // replaced under us: wait for the new one
        if (wall_age_seconds(st) > grace) return rc;                         // nobody is writing this any more: it IS malformed
        usleep(20000);
    }
}

}  // namespace favp
