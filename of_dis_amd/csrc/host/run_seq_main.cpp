// run_OF_INT_seq / run_OF_RGB_seq -- the reference's run_OF_* main (run_dense.cpp:185-431) over MANY frame pairs and
// several GPUs of one node, in the host language of the reference, on top of the C ABI (include/ofdis.h).
//
//   run_OF_INT_seq pairs.txt [--gpus N | --devices d0,d1,..] [--chunk C] [--dry-run 1] [oppoint 1-4 | p1 .. p20]
//
// pairs.txt: one pair per line, "img1 img2 out.flo" (blank lines and lines starting with # are skipped); all images of one
// size.  The parameter block after the options is the single-pair binaries' (README.md:48-88).
//
// The reference has no such tool: its main handles one pair per process.  Frame pairs are independent problems (`initflow`
// is always null, run_dense.cpp:395), so the list is cut into contiguous shares (sizes differ by at most one, earlier
// shares take the remainder -- the partition of of_dis_amd/shard.py: frame_range), one host thread per share, each bound to
// its GPU (ofdis_set_device), with nothing shared on the data path.  A thread streams its share through ONE resident batch
// context of C pairs (default 64), three stages on three threads (decode | device | .flo files) over rotating chunk buffers:
// read C pairs -> upload the 8-bit frames -> padding, pyramid, Sobel on the device
// (ofdis_batch_build_pyramids_u8: run_dense.cpp:130-178,298-344) -> the hot path (ofdis_batch_run: OFClass::OFClass,
// oflow.cpp:184-337) -> x 2^lv_l, bilinear upsample, crop on the device (ofdis_batch_upsample_frames: run_dense.cpp:406-414)
// -> download -> one Middlebury .flo per pair (run_dense.cpp:16-57).  Under the library's default (exact) arithmetic
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
  int failed = 0;   // pairs of this share that could not be processed
  double ms_compute = 0;  // upload .. download of the chunks (device work + PCIe), without image decoding / file writing
  std::string error;      // a failure that ended the share early
};

void frame_range(int total, int rank, int world, int* lo, int* hi) {  // of_dis_amd/shard.py: frame_range
  const int base = total / world, rem = total % world;
  *lo = rank * base + std::min(rank, rem);
  *hi = *lo + base + (rank < rem ? 1 : 0);
}

// One chunk of a share on its way through the three stages: decode (reader thread) -> device (the share's thread) -> .flo
// files (writer thread).  Three buffers rotate, so that a chunk is being decoded and another written while the device works
// on a third: the device stage never waits for the file system unless the file system is the slower side (it is: DESIGN.md 6).
struct Chunk {
  int c0 = 0, m = 0;                // first pair of the chunk (index into the list), pairs in it; m = 0: end of the share
  // [C][h][w][noc] 8-bit frames; slots of unreadable pairs / of a short last chunk keep what the buffer held before (valid
  // images, results never used).  The buffers are not zero-filled up front -- a gigabyte of page faults for nothing -- only the
  // slots that would otherwise go to the device undefined are (init_upto: slots below it have been written at least once)
  std::unique_ptr<uint8_t[]> ha, hb;
  int init_upto = 0;
  std::unique_ptr<float[]> full;    // [C][h][w][2] full-resolution flows
  std::vector<char> ok;
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
 private:
  std::mutex m_;
  std::condition_variable cv_;
  std::deque<Chunk*> q_;
};

void run_share(const std::vector<Pair>& pairs, const ofdis_params& p0, int width_org, int height_org, int chunk, Share* sh) {
  const int n_share = sh->hi - sh->lo;
  if (n_share < 1) return;
  auto bail = [&](const char* what) { sh->error = std::string(what) + ": " + ofdis_last_error(); sh->failed = n_share; };
  if (ofdis_set_device(sh->device) != OFDIS_OK) return bail("ofdis_set_device");
  ofdis_params p = p0;
  p.verbosity = 0;  // (the per-level TIME lines synchronise between stages; the driver prints its own summary)
  const int C = std::min(chunk, n_share);
  ofdis_batch* b = nullptr;
  if (ofdis_batch_create(&b, &p, C) != OFDIS_OK) return bail("ofdis_batch_create");
  const size_t img_bytes = (size_t)width_org * height_org * OFDIS_NOC;
  const size_t flo_floats = (size_t)2 * width_org * height_org;
  void* da = ofdis_dev_alloc(img_bytes * C);
  void* db = ofdis_dev_alloc(img_bytes * C);
  float* dfull = (float*)ofdis_dev_alloc(flo_floats * sizeof(float) * C);
  if (!da || !db || !dfull) {
    bail("ofdis_dev_alloc");
  } else {
    Chunk bufs[3];
    ChunkQueue free_q, ready_q, done_q;
    for (Chunk& c : bufs) {
      c.ha.reset(new uint8_t[img_bytes * C]);
      c.hb.reset(new uint8_t[img_bytes * C]);
      c.full.reset(new float[flo_floats * C]);
      c.ok.assign(C, 0);
      free_q.push(&c);
    }
    std::atomic<int> failed{0};
    std::atomic<bool> stop{false};  // the device stage gave up: the reader stops feeding it
    std::thread reader([&] {  // decode (cv::imread in the reference, run_dense.cpp:208-209)
      for (int c0 = sh->lo; c0 < sh->hi && !stop; c0 += C) {
        Chunk* c = free_q.pop();
        c->c0 = c0;
        c->m = std::min(C, sh->hi - c0);
        for (int k = 0; k < c->m; ++k) {
          const Pair& pr = pairs[c0 + k];
          ofdis_host::Image8 ia, ib;
          std::string err;
          c->ok[k] = ofdis_host::read_image(pr.a, OFDIS_NOC, &ia, &err) && ofdis_host::read_image(pr.b, OFDIS_NOC, &ib, &err);
          if (c->ok[k] && (ia.width != width_org || ia.height != height_org || ib.width != width_org || ib.height != height_org)) {
            c->ok[k] = 0;
            err = pr.a + " / " + pr.b + ": not " + std::to_string(width_org) + "x" + std::to_string(height_org) + " like the first pair";
          }
          if (!c->ok[k]) {
            fprintf(stderr, "%s\n", err.c_str());
            ++failed;
            if (k >= c->init_upto) {  // never written: a defined (black) frame; otherwise the slot keeps its previous content
              memset(c->ha.get() + k * img_bytes, 0, img_bytes);
              memset(c->hb.get() + k * img_bytes, 0, img_bytes);
            }
            continue;
          }
          memcpy(c->ha.get() + k * img_bytes, ia.data.data(), img_bytes);
          memcpy(c->hb.get() + k * img_bytes, ib.data.data(), img_bytes);
        }
        if (c->init_upto < C) {  // the slots of a short chunk that were never written
          const int from = std::max(c->init_upto, c->m);
          if (from < C) {
            memset(c->ha.get() + from * img_bytes, 0, (C - from) * img_bytes);
            memset(c->hb.get() + from * img_bytes, 0, (C - from) * img_bytes);
          }
          c->init_upto = C;
        }
        ready_q.push(c);
      }
      Chunk* end = free_q.pop();
      end->m = 0;
      ready_q.push(end);
    });
    std::thread writer([&] {  // one Middlebury .flo per pair (run_dense.cpp:16-57)
      for (;;) {
        Chunk* c = done_q.pop();
        if (c->m == 0) break;
        for (int k = 0; k < c->m; ++k) {
          if (!c->ok[k]) continue;
          std::string err;
          if (!ofdis_host::write_flo(pairs[c->c0 + k].out, c->full.get() + k * flo_floats, width_org, height_org, &err)) {
            fprintf(stderr, "%s\n", err.c_str());
            ++failed;
          }
        }
        free_q.push(c);
      }
    });
    for (;;) {  // the device stage, on this thread (the one bound to the GPU)
      Chunk* c = ready_q.pop();
      if (c->m == 0) {
        done_q.push(c);
        break;
      }
      if (stop) {  // (drain what the reader had already queued)
        free_q.push(c);
        continue;
      }
      const int m = c->m;
      const double t0 = now_ms();
      int rc = ofdis_memcpy_h2d(da, c->ha.get(), img_bytes * C);
      if (!rc) rc = ofdis_memcpy_h2d(db, c->hb.get(), img_bytes * C);
      if (!rc) rc = ofdis_batch_build_pyramids_u8(b, (const uint8_t*)da, (const uint8_t*)db, width_org, height_org, nullptr);
      for (int attempt = 0; !rc && attempt < 2; ++attempt) {
        rc = ofdis_batch_run(b, nullptr);
        if (!rc) rc = ofdis_batch_upsample_frames(b, 0, m, dfull, width_org, height_org, nullptr);
        if (!rc) rc = ofdis_sync(nullptr);
        if (!rc) rc = ofdis_batch_status(b);
        // A pass that reports itself as failed (a lost hand-over of the cross-CU fused TV variant, small contexts only) is
        // repeated once: the context no longer uses that variant.  ofdis_sync reports such a failure to whoever synchronises the
        // stream of the pass -- with several shares on ONE device (--devices 0,0) that may be another share's thread -- so any
        // OFDIS_ERR_DEVICE gets the one repetition (harmless for a share that was fine; a HIP error proper fails again).
        if (rc == OFDIS_ERR_DEVICE && attempt == 0) {
          fprintf(stderr, "%s\n", ofdis_last_error());
          rc = OFDIS_OK;
          continue;
        }
        break;
      }
      if (!rc) rc = ofdis_memcpy_d2h(c->full.get(), dfull, flo_floats * sizeof(float) * m);
      sh->ms_compute += now_ms() - t0;
      if (rc) {
        sh->error = std::string("chunk: ") + ofdis_last_error();
        failed += sh->hi - c->c0;  // this chunk and everything after it
        stop = true;
        free_q.push(c);
        continue;
      }
      done_q.push(c);
    }
    reader.join();
    writer.join();
    sh->failed = failed;
  }
  if (dfull) ofdis_dev_free(dfull);
  if (da) ofdis_dev_free(da);
  if (db) ofdis_dev_free(db);
  ofdis_batch_destroy(b);
}

}  // namespace

int main(int argc, char** argv) {
  const double t_start = now_ms();
  if (argc < 2) {
    fprintf(stderr, "usage: %s pairs.txt [--gpus N | --devices d0,d1,..] [--chunk C] [--dry-run 1] [oppoint 1-4 | lv_f lv_l maxiter miniter "
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
  int k = 2, chunk = 64;  // (measured: the file system sets the pace; 64-pair chunks keep the three stages busy from the start)
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
  // the geometry of the run: the first pair's size (every other pair is checked against it)
  ofdis_host::Image8 first;
  std::string err;
  if (!ofdis_host::read_image(pairs[0].a, OFDIS_NOC, &first, &err)) {
    fprintf(stderr, "%s\n", err.c_str());
    return 1;
  }
  const int width_org = first.width, height_org = first.height;
  ofdis_params p;
  if (int st = ofdis_host::parse_params(argc, argv, k, width_org, OFDIS_NOC, 1, &p)) return st;
  ofdis_host::pad_size(&p, width_org, height_org);
  const int verbosity = p.verbosity;

  const int R = (int)devices.size();
  std::vector<Share> shares(R);
  std::vector<std::thread> threads;
  const double t0 = now_ms();
  for (int r = 0; r < R; ++r) {
    shares[r].device = devices[r];
    frame_range((int)pairs.size(), r, R, &shares[r].lo, &shares[r].hi);
    threads.emplace_back(run_share, std::cref(pairs), std::cref(p), width_org, height_org, chunk, &shares[r]);
  }
  for (auto& t : threads) t.join();
  const double t_all = now_ms() - t0;
  int failed = 0;
  for (int r = 0; r < R; ++r) {
    failed += shares[r].failed;
    if (!shares[r].error.empty()) fprintf(stderr, "share %d (device %d, pairs %d..%d): %s\n", r, shares[r].device, shares[r].lo, shares[r].hi - 1, shares[r].error.c_str());
    if (verbosity > 1)
      printf("TIME (share %d: device %d, pairs %d..%d, upload+pyramid+flow+upsample+download) (ms): %3g\n", r, shares[r].device,
             shares[r].lo, shares[r].hi - 1, shares[r].ms_compute);
  }
  if (verbosity > 0)
    printf("TIME (%d pairs on %d device share(s), chunk %d, incl. image decoding and .flo writing) (ms): %3g  (%.1f pairs/s; start-up %3g ms)\n",
           (int)pairs.size(), R, chunk, t_all, pairs.size() / (t_all * 1e-3), t0 - t_start);
  return failed ? 1 : 0;
}
