// run_OF_INT_seq / run_OF_RGB_seq -- the reference's run_OF_* main (run_dense.cpp:185-431) over MANY frame pairs and
// several GPUs of one node, in the host language of the reference, on top of the C ABI (include/ofdis.h).
// With -DOFDIS_MODE=2: run_DE_INT_seq / run_DE_RGB_seq, the stereo-depth binaries' counterpart (one displacement channel,
// "img1 img2 out.pfm" per line).
//
//   run_OF_INT_seq pairs.txt [--gpus N | --devices d0,d1,..] [--chunk C] [--depth D] [--dry-run 1] [oppoint 1-4 | p1 .. p20]
//
// pairs.txt: one pair per line, "img1 img2 out.flo" (blank lines and lines starting with # are skipped); all images of one
// size.  The parameter block after the options is the single-pair binaries' (README.md:48-88).
//
// The reference has no such tool: its main handles one pair per process.  Frame pairs are independent problems (`initflow`
// is always null, run_dense.cpp:395), so the list is cut into contiguous shares (sizes differ by at most one, earlier
// shares take the remainder -- the partition of of_dis_amd/shard.py: frame_range), one host thread per share, each bound to
// its GPU (ofdis_set_device), with nothing shared on the data path.  A thread streams its share through resident batch
// contexts of C pairs (default 64), three stages on three threads (decode | device | .flo files) over rotating PINNED chunk
// buffers; the device stage keeps --depth D (default 2) chunks in flight, each on its own slot = context + stream + device
// buffers (round 6: include/ofdis.h version 3), so that one chunk's download (the full-resolution flow is 3.6 MB per
// 1024x436 pair: the link is the bottleneck of this stage) overlaps the next chunk's upload and kernels:
// read C pairs -> upload the 8-bit frames (ofdis_memcpy_h2d_async) -> padding, pyramid, Sobel on the device
// (ofdis_batch_build_pyramids_u8: run_dense.cpp:130-178,298-344) -> the hot path (ofdis_batch_run: OFClass::OFClass,
// oflow.cpp:184-337) -> x 2^lv_l, bilinear upsample, crop on the device (ofdis_batch_upsample_frames: run_dense.cpp:406-414)
// -> download (ofdis_memcpy_d2h_async) -> one Middlebury .flo per pair (run_dense.cpp:16-57).  Under the library's default (exact) arithmetic
// contract every .flo is byte-identical to what the single-pair binary writes for that pair, whatever the chunk size and
// the number of GPUs (tests/test_cli.py::test_sequence_driver_*).
//
// --devices 0,0 puts two shares on one device (how the two-GPU split is tested on a one-GPU box).
// --dry-run 1 prints the partition ("share r: device d pairs lo..hi") and exits without touching a device or a file
// (tests/test_cli.py checks it against of_dis_amd.shard.frame_range on the CPU).
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <fstream>
#include <memory>
#include <mutex>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "cli_params.h"
#include "image_io.h"
#include "ofdis.h"

#ifndef OFDIS_NOC
#define OFDIS_NOC 1
#endif
#ifndef OFDIS_MODE  // the reference's SELECTMODE: 1 optical flow (run_OF_*_seq, .flo), 2 stereo depth (run_DE_*_seq, .pfm)
#define OFDIS_MODE 1
#endif
#define OFDIS_NCH (OFDIS_MODE == 2 ? 1 : 2)

namespace {

double now_ms() {
  struct timeval tv;
  gettimeofday(&tv, nullptr);
  return tv.tv_sec * 1000.0 + tv.tv_usec / 1000.0;
}

struct Pair {
  std::string a, b, out;
};

struct Share {      // one host thread = one contiguous block of the list on one device
  int device = 0;
  int lo = 0, hi = 0;
  int failed = 0;   // pairs of this share whose .flo was NOT written (unreadable, failed on the device, or unwritable)
  double ms_compute = 0;  // the device stage's busy time: first upload .. last download of its chunks (device work + PCIe,
                          // chunks overlapping), without the time it waited for the decoder / the writer
  std::string pci;        // PCI bus id of the device the share ran on
  int chunks_done = 0, chunk_pairs = 0;
  std::string error;      // a failure that ended the share early
};

void frame_range(int total, int rank, int world, int* lo, int* hi) {  // of_dis_amd/shard.py: frame_range
  const int base = total / world, rem = total % world;
  *lo = rank * base + std::min(rank, rem);
  *hi = *lo + base + (rank < rem ? 1 : 0);
}

// The pairs of a chunk are decoded / written by `threads` workers side by side (--io-threads: the file system and the PNM / PNG
// decoder, not the device, set the pace of a run -- one worker writes ~0.7 k .flo files of 1024x436 per second).
template <typename F>
void parallel_for(int n, int threads, F&& f) {
  threads = std::max(1, std::min(threads, n));
  if (threads == 1) {
    for (int k = 0; k < n; ++k) f(k);
    return;
  }
  std::atomic<int> next{0};
  std::vector<std::thread> pool;
  for (int t = 0; t < threads; ++t)
    pool.emplace_back([&] {
      for (int k = next++; k < n; k = next++) f(k);
    });
  for (auto& t : pool) t.join();
}

// One chunk of a share on its way through the three stages: decode (reader thread) -> device (the share's thread) -> .flo
// files (writer thread).  Three buffers rotate, so that a chunk is being decoded and another written while the device works
// on a third: the device stage never waits for the file system unless the file system is the slower side (it is: DESIGN.md 6).
struct Chunk {
  int c0 = 0, m = 0;                // first pair of the chunk (index into the list), pairs in it; m = 0: end of the share
  // [C][h][w][noc] 8-bit frames; slots of unreadable pairs / of a short last chunk keep what the buffer held before (valid
  // images, results never used).  The buffers are not zero-filled up front -- a gigabyte of page faults for nothing -- only the
  // slots that would otherwise go to the device undefined are (init_upto: slots below it have been written at least once)
  // (pinned: ofdis_host_alloc -- the asynchronous copies are DMAs only from / to page-locked memory)
  uint8_t *ha = nullptr, *hb = nullptr;
  int init_upto = 0;
  float* full = nullptr;            // [C][h][w][2] full-resolution flows
  std::vector<char> ok;
};
// One slot of the device stage: a resident context with its own compute stream and device buffers.  The copies do NOT run on
// the slot's stream: all uploads of a share go, in order, to one upload stream and all downloads to one download stream (the
// link has one direction each; two slots that each ran upload - kernels - download on a stream of their own would fall into
// lock step -- both uploading, then both downloading -- and the directions would never overlap: tools/link_probe.py), tied
// to the slot's kernels by events.
struct Slot {
  ofdis_batch* b = nullptr;
  void* stream = nullptr;           // pyramids, the hot path, upsample
  void *da = nullptr, *db = nullptr;
  float* dfull = nullptr;
  void *ev_up = nullptr, *ev_done = nullptr, *ev_down = nullptr;  // upload complete / kernels complete / download complete
  Chunk* chunk = nullptr;           // the chunk in flight on this slot, or null
};
class ChunkQueue {  // a blocking FIFO of chunk pointers
 public:
  void push(Chunk* c) {
    { std::lock_guard<std::mutex> l(m_); q_.push_back(c); }
    cv_.notify_one();
  }
  Chunk* pop() {
    std::unique_lock<std::mutex> l(m_);
    cv_.wait(l, [&] { return !q_.empty(); });
    Chunk* c = q_.front();
    q_.pop_front();
    return c;
  }
  Chunk* try_pop() {  // null when the queue is empty
    std::lock_guard<std::mutex> l(m_);
    if (q_.empty()) return nullptr;
    Chunk* c = q_.front();
    q_.pop_front();
    return c;
  }
 private:
  std::mutex m_;
  std::condition_variable cv_;
  std::deque<Chunk*> q_;
};

// device_bench > 0 (--device-bench N, a measurement mode): the first chunk of the share is decoded once and pushed through the
// device stage N times, no .flo is written -- what the device stage (link + kernels) sustains when neither the decoder nor
// the file system holds it back (tools/seq_probe.py).
void run_share(const std::vector<Pair>& pairs, const ofdis_params& p0, int width_org, int height_org, int chunk, int depth,
               int device_bench, int io_threads, Share* sh) {
  const int n_share = sh->hi - sh->lo;
  if (n_share < 1) return;
  auto bail = [&](const char* what) { sh->error = std::string(what) + ": " + ofdis_last_error(); sh->failed = n_share; };
  if (ofdis_set_device(sh->device) != OFDIS_OK) return bail("ofdis_set_device");
  {
    char id[32] = {0};
    if (ofdis_device_pci_bus_id(sh->device, id, sizeof(id)) == OFDIS_OK) sh->pci = id;
  }
  ofdis_params p = p0;
  p.verbosity = 0;  // (the per-level TIME lines synchronise between stages; the driver prints its own summary)
  const int C = std::min(chunk, n_share);
  sh->chunk_pairs = C;
  const int D = std::max(1, std::min(depth, device_bench > 0 ? device_bench : (n_share + C - 1) / C));  // slots in flight on the device
  const size_t img_bytes = (size_t)width_org * height_org * OFDIS_NOC;
  const size_t flo_floats = (size_t)OFDIS_NCH * width_org * height_org;
  std::vector<Slot> slots(D);
  // chunk buffers: one being decoded, D on the device, one being written
  std::vector<Chunk> bufs(D + 2);
  bool ok = true;
  void* s_in = ofdis_stream_create();   // every upload of this share, in order
  void* s_out = ofdis_stream_create();  // every download of this share, in order
  if (!s_in || !s_out) { bail("ofdis_stream_create"); ok = false; }
  for (Slot& sl : slots) {
    if (!ok) break;
    if (ofdis_batch_create(&sl.b, &p, C) != OFDIS_OK) { bail("ofdis_batch_create"); ok = false; break; }
    sl.stream = ofdis_stream_create();
    sl.ev_up = ofdis_event_create();
    sl.ev_done = ofdis_event_create();
    sl.ev_down = ofdis_event_create();
    if (!sl.ev_up || !sl.ev_done || !sl.ev_down) { bail("ofdis_event_create"); ok = false; break; }
    sl.da = ofdis_dev_alloc(img_bytes * C);
    sl.db = ofdis_dev_alloc(img_bytes * C);
    sl.dfull = (float*)ofdis_dev_alloc(flo_floats * sizeof(float) * C);
    if (!sl.stream || !sl.da || !sl.db || !sl.dfull) { bail("slot allocation"); ok = false; break; }
  }
  // (page-locking memory costs ~0.25 ms per MB, a 64-pair chunk of 1024x436 frames is 285 MB: the buffers are allocated by the
  // reader thread when it first needs them, beside the device work on the chunks before)
  std::atomic<bool> alloc_failed{false};
  auto chunk_alloc = [&](Chunk& c) {
    if (c.ha) return true;
    c.ha = (uint8_t*)ofdis_host_alloc(img_bytes * C);
    c.hb = (uint8_t*)ofdis_host_alloc(img_bytes * C);
    c.full = (float*)ofdis_host_alloc(flo_floats * sizeof(float) * C);
    c.ok.assign(C, 0);
    return c.ha && c.hb && c.full;
  };
  const int n_chunks = device_bench > 0 ? device_bench : (n_share + C - 1) / C;
  if (ok) {
    ChunkQueue free_q, ready_q, done_q;
    for (Chunk& c : bufs) free_q.push(&c);
    std::atomic<int> written{0};    // .flo files of this share that exist now
    std::atomic<bool> stop{false};  // the device stage gave up: the reader stops feeding it
    std::thread reader([&] {  // decode (cv::imread in the reference, run_dense.cpp:208-209)
      for (int ci = 0; ci < n_chunks && !stop; ++ci) {
        const int c0 = device_bench > 0 ? sh->lo : sh->lo + ci * C;
        Chunk* c = free_q.pop();
        if (!chunk_alloc(*c)) {
          fprintf(stderr, "ofdis_host_alloc: %s\n", ofdis_last_error());
          alloc_failed = true;
          free_q.push(c);
          break;
        }
        c->c0 = c0;
        c->m = std::min(C, sh->hi - c0);
        if (device_bench > 0 && c->init_upto >= C) {  // (measurement mode: this buffer already holds the decoded first chunk)
          ready_q.push(c);
          continue;
        }
        parallel_for(c->m, io_threads, [&](int k) {
          const Pair& pr = pairs[c0 + k];
          ofdis_host::Image8 ia, ib;
          std::string err;
          c->ok[k] = ofdis_host::read_image(pr.a, OFDIS_NOC, &ia, &err) && ofdis_host::read_image(pr.b, OFDIS_NOC, &ib, &err);
          if (c->ok[k] && (ia.width != width_org || ia.height != height_org || ib.width != width_org || ib.height != height_org)) {
            c->ok[k] = 0;
            err = pr.a + " / " + pr.b + ": not " + std::to_string(width_org) + "x" + std::to_string(height_org) + " like the first readable pair";
          }
          if (!c->ok[k]) {
            fprintf(stderr, "%s\n", err.c_str());
            if (k >= c->init_upto) {  // never written: a defined (black) frame; otherwise the slot keeps its previous content
              memset(c->ha + k * img_bytes, 0, img_bytes);
              memset(c->hb + k * img_bytes, 0, img_bytes);
            }
            return;
          }
          memcpy(c->ha + k * img_bytes, ia.data.data(), img_bytes);
          memcpy(c->hb + k * img_bytes, ib.data.data(), img_bytes);
        });
        if (c->init_upto < C) {  // the slots of a short chunk that were never written
          const int from = std::max(c->init_upto, c->m);
          if (from < C) {
            memset(c->ha + from * img_bytes, 0, (C - from) * img_bytes);
            memset(c->hb + from * img_bytes, 0, (C - from) * img_bytes);
          }
          c->init_upto = C;
        }
        ready_q.push(c);
      }
      Chunk* end = free_q.pop();
      end->m = 0;
      ready_q.push(end);
    });
    const bool discard = device_bench > 0;
    std::thread writer([&] {  // one Middlebury .flo per pair (run_dense.cpp:16-57)
      for (;;) {
        Chunk* c = done_q.pop();
        if (c->m == 0) break;
        parallel_for(discard ? 0 : c->m, io_threads, [&](int k) {
          if (!c->ok[k]) return;
          std::string err;
#if OFDIS_MODE == 2
          if (!ofdis_host::write_pfm(pairs[c->c0 + k].out, c->full + k * flo_floats, width_org, height_org, &err))
#else
          if (!ofdis_host::write_flo(pairs[c->c0 + k].out, c->full + k * flo_floats, width_org, height_org, &err))
#endif
            fprintf(stderr, "%s\n", err.c_str());
          else
            ++written;
        });
        free_q.push(c);
      }
    });
    // ---- the device stage, on this thread (the one bound to the GPU)
    // upload (upload stream) -> pyramids, the hot path, upsample (the slot's stream) -> download (download stream)
    auto enqueue = [&](Slot& sl, Chunk* c, bool with_upload) -> int {
      int rc = OFDIS_OK;
      if (with_upload) {
        // (the slot's device buffers are free: retire() waited for the download that followed the kernels that last read them)
        rc = ofdis_memcpy_h2d_async(sl.da, c->ha, img_bytes * C, s_in);
        if (!rc) rc = ofdis_memcpy_h2d_async(sl.db, c->hb, img_bytes * C, s_in);
        if (!rc) rc = ofdis_event_record(sl.ev_up, s_in);
        if (!rc) rc = ofdis_stream_wait_event(sl.stream, sl.ev_up);
        if (!rc) rc = ofdis_batch_build_pyramids_u8(sl.b, (const uint8_t*)sl.da, (const uint8_t*)sl.db, width_org, height_org, sl.stream);
      }
      if (!rc) rc = ofdis_batch_run(sl.b, sl.stream);
      if (!rc) rc = ofdis_batch_upsample_frames(sl.b, 0, c->m, sl.dfull, width_org, height_org, sl.stream);
      if (!rc) rc = ofdis_event_record(sl.ev_done, sl.stream);
      if (!rc) rc = ofdis_stream_wait_event(s_out, sl.ev_done);
      if (!rc) rc = ofdis_memcpy_d2h_async(c->full, sl.dfull, flo_floats * sizeof(float) * c->m, s_out);
      if (!rc) rc = ofdis_event_record(sl.ev_down, s_out);
      return rc;
    };
    // wait for the slot's chunk (its download) and hand it to the writer.  A pass that reports itself as failed (a lost
    // hand-over of the cross-CU fused TV variant, small contexts only: ofdis_batch_status of THIS slot's context, whose pass
    // the download followed) is repeated once -- the pyramids are still resident and the context no longer uses that variant.
    auto retire = [&](Slot& sl) -> int {
      Chunk* c = sl.chunk;
      if (!c) return OFDIS_OK;
      int rc = ofdis_event_sync(sl.ev_down);
      if (!rc) rc = ofdis_batch_status(sl.b);
      if (rc == OFDIS_ERR_DEVICE) {
        fprintf(stderr, "%s\n", ofdis_last_error());
        rc = enqueue(sl, c, false);
        if (!rc) rc = ofdis_event_sync(sl.ev_down);
        if (!rc) rc = ofdis_batch_status(sl.b);
      }
      sl.chunk = nullptr;
      if (rc) {
        free_q.push(c);
        return rc;
      }
      done_q.push(c);
      return OFDIS_OK;
    };
    double busy_since = 0;  // start of the current busy interval (some slot has a chunk), 0 = idle
    // (measurement mode: the first D + 2 chunks page-lock and fill the chunk buffers -- not counted)
    const int bench_skip = device_bench > D + 2 ? D + 2 : 0;
    int chunks_seen = 0;
    auto oldest_in_flight = [&](int from) -> Slot* {
      for (int k = 0; k < D; ++k)
        if (slots[(from + k) % D].chunk) return &slots[(from + k) % D];
      return nullptr;
    };
    auto note_error = [&] { if (sh->error.empty()) sh->error = std::string("chunk: ") + ofdis_last_error(); };
    int next = 0;  // slots are used round-robin: the oldest chunk in flight is the first occupied slot from `next` on
    for (;;) {
      // nothing decoded yet: rather than sitting on finished chunks, hand the oldest one in flight to the writer
      Chunk* c = ready_q.try_pop();
      if (!c) {
        if (Slot* o = oldest_in_flight(next)) {
          if (retire(*o)) { note_error(); stop = true; }
          if (!oldest_in_flight(next) && busy_since != 0) { sh->ms_compute += now_ms() - busy_since; busy_since = 0; }
          continue;
        }
        c = ready_q.pop();  // (the device is idle: this wait is the decoder's time, not the device stage's)
      }
      if (c->m == 0) {  // end of the share: retire what is in flight, oldest first
        while (Slot* o = oldest_in_flight(next))
          if (retire(*o)) note_error();
        if (busy_since != 0) { sh->ms_compute += now_ms() - busy_since; busy_since = 0; }
        done_q.push(c);
        break;
      }
      if (stop) {  // (drain what the reader had already queued)
        free_q.push(c);
        continue;
      }
      Slot& sl = slots[next];
      next = (next + 1) % D;
      int rc = retire(sl);  // the slot's previous chunk (the D - 1 younger chunks stay in flight meanwhile)
      if (++chunks_seen == bench_skip + 1 && bench_skip) { sh->ms_compute = 0; busy_since = 0; }  // the measured part starts here
      if (busy_since == 0) busy_since = now_ms();
      if (!rc) {
        rc = enqueue(sl, c, true);
        if (!rc) sl.chunk = c;
      }
      if (rc) {  // this chunk and everything after it stay unwritten; what is in flight finishes and is written
        note_error();
        stop = true;
        if (!sl.chunk) {
          (void)ofdis_sync(s_in);  // (copies of a half-enqueued chunk may still be reading its buffers)
          (void)ofdis_sync(s_out);
          free_q.push(c);
        }
      }
    }
    reader.join();
    writer.join();
    sh->failed = discard ? 0 : n_share - written;  // as a set: the pairs of this share without a .flo (never more than the share)
    if (alloc_failed && sh->error.empty()) sh->error = "pinned host memory";
    sh->chunks_done = n_chunks - bench_skip;
  }
  for (Chunk& c : bufs) {
    ofdis_host_free(c.ha);
    ofdis_host_free(c.hb);
    ofdis_host_free(c.full);
  }
  if (s_in) (void)ofdis_sync(s_in);
  if (s_out) (void)ofdis_sync(s_out);
  for (Slot& sl : slots) {
    if (sl.stream) (void)ofdis_sync(sl.stream);
    ofdis_event_destroy(sl.ev_up);
    ofdis_event_destroy(sl.ev_done);
    ofdis_event_destroy(sl.ev_down);
    if (sl.dfull) ofdis_dev_free(sl.dfull);
    if (sl.da) ofdis_dev_free(sl.da);
    if (sl.db) ofdis_dev_free(sl.db);
    ofdis_batch_destroy(sl.b);
    ofdis_stream_destroy(sl.stream);
  }
  ofdis_stream_destroy(s_in);
  ofdis_stream_destroy(s_out);
}

}  // namespace

int main(int argc, char** argv) {
  const double t_start = now_ms();
  if (argc < 2) {
    fprintf(stderr, "usage: %s pairs.txt [--gpus N | --devices d0,d1,..] [--chunk C] [--depth D] [--size W H] [--device-bench N] [--io-threads T] [--dry-run 1] [oppoint 1-4 | lv_f lv_l maxiter miniter "
                    "mindprate mindrrate minimgerr patchsz poverl usefbcon patnorm costfct usetvref tv_alpha tv_gamma tv_delta "
                    "tv_innerit tv_solverit tv_sor verbosity]\n  pairs.txt: one \"img1 img2 out.flo\" per line\n", argv[0]);
    return 2;
  }
  std::vector<Pair> pairs;
  {
    std::ifstream f(argv[1]);
    if (!f) {
      fprintf(stderr, "cannot read %s\n", argv[1]);
      return 1;
    }
    std::string line;
    while (std::getline(f, line)) {
      std::istringstream ss(line);
      Pair pr;
      if (!(ss >> pr.a) || pr.a[0] == '#') continue;
      if (!(ss >> pr.b >> pr.out)) {
        fprintf(stderr, "%s: expected \"img1 img2 out.flo\", got \"%s\"\n", argv[1], line.c_str());
        return 2;
      }
      pairs.push_back(pr);
    }
  }
  if (pairs.empty()) {
    fprintf(stderr, "%s lists no pairs\n", argv[1]);
    return 1;
  }
  int k = 2, chunk = 16;  // (measured, round 6: with the copies on their own streams the device stage is link-bound from 16-pair
                          // chunks on -- 13.7 k pairs/s against 11.5 k with 64-pair chunks over a run of 8192 pairs, whose
                          // four 285 MB chunk buffers take 0.3 s to page-lock; small chunks also start the three stages sooner)
  int depth = 2;          // chunks in flight on the device (slots): 2 = one chunk's download overlaps the next one's upload + kernels
  int size_w = 0, size_h = 0;  // --size W H: the geometry of the run (default: the first readable pair's)
  int device_bench = 0;        // --device-bench N: measurement mode (run_share)
  int io_threads = 0;          // --io-threads T: decoder / writer workers per share (0 = hardware threads / (2 x shares), 1..16)
  bool dry_run = false;
  std::vector<int> devices;
  while (k < argc && argv[k][0] == '-' && argv[k][1] == '-') {
    const std::string opt = argv[k];
    if (k + 1 >= argc) {
      fprintf(stderr, "%s needs a value\n", opt.c_str());
      return 2;
    }
    const char* val = argv[k + 1];
    if (opt == "--gpus") {
      devices.clear();
      for (int d = 0; d < atoi(val); ++d) devices.push_back(d);
    } else if (opt == "--devices") {
      devices.clear();
      std::istringstream ss(val);
      std::string tok;
      while (std::getline(ss, tok, ',')) devices.push_back(atoi(tok.c_str()));
    } else if (opt == "--chunk") {
      chunk = atoi(val);
    } else if (opt == "--depth") {
      depth = atoi(val);
    } else if (opt == "--device-bench") {
      device_bench = atoi(val);
    } else if (opt == "--io-threads") {
      io_threads = atoi(val);
    } else if (opt == "--size") {
      if (k + 2 >= argc) {
        fprintf(stderr, "--size needs W H\n");
        return 2;
      }
      size_w = atoi(val);
      size_h = atoi(argv[k + 2]);
      ++k;
    } else if (opt == "--dry-run") {
      dry_run = atoi(val) != 0;
    } else {
      fprintf(stderr, "unknown option %s\n", opt.c_str());
      return 2;
    }
    k += 2;
  }
  if (devices.empty()) devices.push_back(0);
  if (dry_run) {  // the partition only
    for (int r = 0; r < (int)devices.size(); ++r) {
      int lo, hi;
      frame_range((int)pairs.size(), r, (int)devices.size(), &lo, &hi);
      printf("share %d: device %d pairs %d..%d\n", r, devices[r], lo, hi - 1);
    }
    return 0;
  }
  const int ndev = ofdis_device_count();
  for (int d : devices)
    if (d < 0 || d >= ndev) {
      fprintf(stderr, "device %d requested, %d HIP device(s) visible\n", d, ndev);
      return 1;
    }
  if (chunk < 1 || chunk > 65535) {
    fprintf(stderr, "--chunk must be 1..65535\n");
    return 2;
  }
  if (depth < 1 || depth > 8) {
    fprintf(stderr, "--depth must be 1..8\n");
    return 2;
  }
  // the geometry of the run: --size, else the size of the first READABLE first image (an unreadable pair is reported by its
  // share like any other and does not stop the pairs after it); every pair is checked against it
  int width_org = size_w, height_org = size_h;
  if (width_org < 1 || height_org < 1) {
    std::string err;
    for (size_t i = 0; i < pairs.size() && width_org < 1; ++i) {
      ofdis_host::Image8 first;
      if (ofdis_host::read_image(pairs[i].a, OFDIS_NOC, &first, &err)) {
        width_org = first.width;
        height_org = first.height;
      }
    }
    if (width_org < 1) {
      fprintf(stderr, "no readable image in %s (last error: %s)\n", argv[1], err.c_str());
      return 1;
    }
  }
  ofdis_params p;
  if (int st = ofdis_host::parse_params(argc, argv, k, width_org, OFDIS_NOC, OFDIS_MODE, &p)) return st;
  ofdis_host::pad_size(&p, width_org, height_org);
  const int verbosity = p.verbosity;

  const int R = (int)devices.size();
  if (io_threads < 1) io_threads = std::max(1, std::min(16, (int)std::thread::hardware_concurrency() / (2 * R)));
  std::vector<Share> shares(R);
  std::vector<std::thread> threads;
  const double t0 = now_ms();
  for (int r = 0; r < R; ++r) {
    shares[r].device = devices[r];
    frame_range((int)pairs.size(), r, R, &shares[r].lo, &shares[r].hi);
    threads.emplace_back(run_share, std::cref(pairs), std::cref(p), width_org, height_org, chunk, depth, device_bench, io_threads, &shares[r]);
  }
  for (auto& t : threads) t.join();
  const double t_all = now_ms() - t0;
  int failed = 0;
  for (int r = 0; r < R; ++r) {
    failed += shares[r].failed;
    if (!shares[r].error.empty()) fprintf(stderr, "share %d (device %d, pairs %d..%d): %s\n", r, shares[r].device, shares[r].lo, shares[r].hi - 1, shares[r].error.c_str());
    if (verbosity > 1)
      printf("TIME (share %d: device %d [%s], pairs %d..%d, upload+pyramid+flow+upsample+download, %d chunk(s) in flight) (ms): %3g\n", r,
             shares[r].device, shares[r].pci.c_str(), shares[r].lo, shares[r].hi - 1, depth, shares[r].ms_compute);
  }
  if (device_bench > 0)
    for (int r = 0; r < R; ++r)
      printf("DEVICE STAGE (share %d: %d chunks of %d pairs, %d in flight): %.1f pairs/s (upload of the 8-bit frames, pyramids, flow, upsample, "
             "download of the full-resolution flow; no decoding, no .flo)\n", r, shares[r].chunks_done, shares[r].chunk_pairs, depth,
             shares[r].ms_compute > 0 ? shares[r].chunks_done * (double)shares[r].chunk_pairs / (shares[r].ms_compute * 1e-3) : 0.0);
  if (verbosity > 0)
    printf("TIME (%d pairs on %d device share(s), chunk %d, incl. image decoding and .flo writing by %d worker(s) each per share) (ms): %3g  (%.1f pairs/s; start-up %3g ms)\n",
           (int)pairs.size(), R, chunk, io_threads, t_all, pairs.size() / (t_all * 1e-3), t0 - t_start);
  return failed ? 1 : 0;
}
