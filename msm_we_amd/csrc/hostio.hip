// hostio.hip -- the host side of staging in native threads (hamsm.h, ABI 8): byte ranges of the source (arrays in memory, or
// ranges of raw files) into the page-locked upload slab.  No device work here; it lives in the library so that the copies do not
// queue behind the interpreter lock (a Python pool of 12 readers moves 1 GB in 26-50 ms whatever the job size: ~60 us of dispatch
// per job; memcpy with 12 threads needs 8 ms).  Replaces `_data.py:254-320, 531-618` (per-iteration reads, per-segment appends).
#include <errno.h>
#include <unistd.h>

#include <atomic>
#include <thread>

#include "common.h"

namespace {
constexpr int64_t PIECE = 1 << 20;

struct Piece {
  int64_t part, off, len;
};

// run f(piece) over all pieces on `threads` threads (the calling thread is one of them); first error wins
template <typename F>
int run_pieces(hamsm_ctx *c, int64_t n_parts, const int64_t *bytes, int threads, const char *what, F f) {
  std::vector<Piece> pieces;
  for (int64_t i = 0; i < n_parts; i++) {
    if (bytes[i] < 0) return hamsm_fail(c, HAMSM_EINVAL, "%s: part %lld has a negative size", what, (long long)i);
    for (int64_t o = 0; o < bytes[i]; o += PIECE) pieces.push_back({i, o, bytes[i] - o < PIECE ? bytes[i] - o : PIECE});
  }
  if (pieces.empty()) return HAMSM_OK;
  if (threads < 1) threads = 1;
  if (threads > 64) threads = 64;
  if ((int64_t)threads > (int64_t)pieces.size()) threads = (int)pieces.size();
  std::atomic<int64_t> next{0};
  std::atomic<int> err{0};
  std::atomic<int64_t> bad_part{-1};
  auto work = [&]() {
    for (;;) {
      const int64_t k = next.fetch_add(1, std::memory_order_relaxed);
      if (k >= (int64_t)pieces.size() || err.load(std::memory_order_relaxed)) return;
      const int e = f(pieces[k]);
      if (e) {
        int zero = 0;
        if (err.compare_exchange_strong(zero, e)) bad_part.store(pieces[k].part);
      }
    }
  };
  std::vector<std::thread> pool;
  pool.reserve(threads - 1);
  for (int t = 1; t < threads; t++) {
    try {
      pool.emplace_back(work);
    } catch (...) {  // (no more threads to be had: the ones that exist, this one included, take all pieces)
      break;
    }
  }
  work();
  for (auto &t : pool) t.join();
  if (err.load()) return hamsm_fail(c, HAMSM_EINVAL, "%s: part %lld of %lld: %s", what, (long long)bad_part.load(), (long long)n_parts,
                                    err.load() == -1 ? "short read (the file ends inside the range)" : strerror(err.load()));
  return HAMSM_OK;
}
}  // namespace

extern "C" int hamsm_host_copy_parts(hamsm_ctx *c, int64_t n_parts, const void *const *h_src, void *const *h_dst, const int64_t *bytes,
                                     int threads) {
  CTX(c);
  if (n_parts < 0 || (n_parts > 0 && (!h_src || !h_dst || !bytes))) return hamsm_fail(c, HAMSM_EINVAL, "host_copy_parts: null table");
  for (int64_t i = 0; i < n_parts; i++)
    if (bytes[i] > 0 && (!h_src[i] || !h_dst[i])) return hamsm_fail(c, HAMSM_EINVAL, "host_copy_parts: part %lld has a null pointer", (long long)i);
  return run_pieces(c, n_parts, bytes, threads, "host_copy_parts", [&](const Piece &p) {
    memcpy((char *)h_dst[p.part] + p.off, (const char *)h_src[p.part] + p.off, (size_t)p.len);
    return 0;
  });
}

extern "C" int hamsm_host_read_parts(hamsm_ctx *c, int64_t n_parts, const int32_t *fd, const int64_t *file_off, void *const *h_dst,
                                     const int64_t *bytes, int threads) {
  CTX(c);
  if (n_parts < 0 || (n_parts > 0 && (!fd || !file_off || !h_dst || !bytes))) return hamsm_fail(c, HAMSM_EINVAL, "host_read_parts: null table");
  for (int64_t i = 0; i < n_parts; i++)
    if (bytes[i] > 0 && (!h_dst[i] || fd[i] < 0 || file_off[i] < 0))
      return hamsm_fail(c, HAMSM_EINVAL, "host_read_parts: part %lld has a null pointer, a closed descriptor or a negative offset", (long long)i);
  return run_pieces(c, n_parts, bytes, threads, "host_read_parts", [&](const Piece &p) {
    char *dst = (char *)h_dst[p.part] + p.off;
    int64_t done = 0;
    while (done < p.len) {
      const ssize_t r = pread(fd[p.part], dst + done, (size_t)(p.len - done), (off_t)(file_off[p.part] + p.off + done));
      if (r < 0) {
        if (errno == EINTR) continue;
        return errno ? errno : EIO;
      }
      if (r == 0) return -1;
      done += r;
    }
    return 0;
  });
}
