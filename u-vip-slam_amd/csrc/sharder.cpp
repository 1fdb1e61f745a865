// Multi-GPU form of the extractor + consecutive-frame matcher (the single extractor call site it feeds is
// src/Tracking.cc:946).  A frame's extraction depends on nothing but its own pixels, so a job of `total` frames is cut into one
// contiguous block per shard; every shard has its own extractor + matcher handle on its device and one host thread that streams
// the block through the two pipeline lanes in chunks (upload of chunk k+1 under the kernels of chunk k).  There is no collective:
// every chunk's results are copied by hipMemcpyAsync straight to their final place -- offset frame * cap of ONE set of page-locked
// host arrays shared by all shards -- and that is the gather.  Matching pair p = (frame p, frame p+1) needs frame p+1's
// descriptors: a chunk therefore extracts one frame more than it owns (its halo: the first frame of the next chunk, which at a
// shard's end belongs to the neighbouring shard) instead of exchanging descriptors between devices; +1/chunk extra work.
//
// Every local shard has a persistent host thread with a queue of jobs: uvo_sharder_submit() hands a job to the queues and returns,
// uvo_sharder_wait() blocks until its results are in host memory, uvo_sharder_run() is the two together.  A shard's thread keeps at
// most two chunks in flight (one per pipeline lane) and does NOT drain them at the end of a job: the first chunk of the next job is
// uploaded under the kernels of the last chunk of this one, so a stream of jobs runs at the steady-state rate of the lanes.
//
// Shards whose device is UVO_SHARD_REMOTE are owned by another process (one process per GPU under a launcher such as
// torch.distributed.run): the same plan, the same offsets, each process runs its own shards, and the output arrays are one shared
// mapping registered with uvo_host_register() in every process.
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include "common.hpp"

extern "C" int uvo_extract_batch_submit_internal(uvo_extractor* h, int batch, int n_download, const uint8_t* imgs, int width, int height,
                                                  ptrdiff_t stride, ptrdiff_t frame_stride, uvo_keypoint* out_kp, uint8_t* out_desc, int cap,
                                                  int32_t* n_out, int* ticket, hipEvent_t after_kernels, const uint8_t** d_desc,
                                                  const int32_t** d_n);
extern "C" hipStream_t uvo_matcher_stream_internal(uvo_matcher* m);
extern "C" int uvo_extractor_next_lane_internal(const uvo_extractor* h);
extern "C" int uvo_extract_batch_done_internal(uvo_extractor* h, int ticket);

using namespace uvo;

namespace {

struct Shard {
  int device = UVO_SHARD_REMOTE;
  uvo_extractor* ex = nullptr;
  uvo_matcher* mt = nullptr;
  // per pipeline lane: knn-2 rows of the chunk in HBM, and the events that order the two handles' streams
  int32_t *d_idx0[2] = {nullptr, nullptr}, *d_idx1[2] = {nullptr, nullptr};
  uint16_t *d_d0[2] = {nullptr, nullptr}, *d_d1[2] = {nullptr, nullptr};
  hipEvent_t kernels_done[2] = {nullptr, nullptr};  // extractor lane: descriptors of the chunk are final
  hipEvent_t rows_sent[2] = {nullptr, nullptr};     // matcher stream: the chunk's rows are on their way to the host
  std::thread worker;
  int32_t numa_node = -1;  // of the shard's device, once its thread has bound itself (uvo_host_bind_near_device)
};

struct RunArgs {
  const uint8_t* imgs;
  int imgs_first_frame, total, width, height;
  ptrdiff_t stride, frame_stride;
  uvo_keypoint* out_kp;
  uint8_t* out_desc;
  int cap;
  int32_t* n_out;
  int32_t *idx0, *idx1;
  uint16_t *d0, *d1;
};

struct Job {
  int id = 0;
  RunArgs a;
  int remaining = 0;  // local shards that have not delivered yet
  int rc = UVO_OK;
  std::string err;
};

}  // namespace

struct uvo_sharder {
  uvo_sharder_cfg cfg;
  std::vector<Shard> shards;
  int dcap = 0;  // keypoints a frame can return (uvo_extractor_max_keypoints)
  // job queue shared by the shard threads (every local shard works through the jobs in submission order)
  std::mutex mu;
  std::condition_variable cv_work, cv_done;
  std::deque<std::shared_ptr<Job>> jobs;  // submitted, not yet waited for
  int next_id = 1;
  bool stopping = false;
};

namespace {
void shard_worker(uvo_sharder* s, int shard_index);
}

extern "C" {

int uvo_shard_plan_make(int total_frames, int n_shards, int shard, int chunk_frames, uvo_shard_plan* out) {
  if (!out || total_frames < 0 || n_shards < 1 || shard < 0 || shard >= n_shards || chunk_frames < 1) return fail(UVO_E_BADARG, "bad shard plan request");
  const int base = total_frames / n_shards, rem = total_frames % n_shards;
  out->first_frame = shard * base + std::min(shard, rem);
  out->n_frames = base + (shard < rem ? 1 : 0);
  const int end = out->first_frame + out->n_frames;
  // pair p = (frame p, frame p + 1), p in [0, total - 1): a shard matches the pairs whose first frame it owns
  out->first_pair = out->first_frame;
  out->n_pairs = std::max(std::min(end, total_frames - 1) - out->first_frame, 0);
  out->halo_frame = (out->n_frames > 0 && end < total_frames) ? end : -1;
  out->n_chunks = (out->n_frames + chunk_frames - 1) / chunk_frames;
  return UVO_OK;
}

void uvo_sharder_destroy(uvo_sharder* s) {
  if (!s) return;
  {
    std::lock_guard<std::mutex> lk(s->mu);
    s->stopping = true;
  }
  s->cv_work.notify_all();
  for (Shard& sh : s->shards)
    if (sh.worker.joinable()) sh.worker.join();
  for (Shard& sh : s->shards) {
    if (sh.device == UVO_SHARD_REMOTE) continue;
    (void)hipSetDevice(sh.device);
    if (sh.ex) (void)uvo_extractor_synchronize(sh.ex);
    if (sh.mt) (void)uvo_matcher_synchronize(sh.mt);
    for (int l = 0; l < 2; ++l) {
      void* p[] = {sh.d_idx0[l], sh.d_idx1[l], sh.d_d0[l], sh.d_d1[l]};
      for (void* q : p)
        if (q) (void)hipFree(q);
      if (sh.kernels_done[l]) (void)hipEventDestroy(sh.kernels_done[l]);
      if (sh.rows_sent[l]) (void)hipEventDestroy(sh.rows_sent[l]);
    }
    if (sh.mt) uvo_matcher_destroy(sh.mt);
    if (sh.ex) uvo_extractor_destroy(sh.ex);
  }
  delete s;
}

int uvo_sharder_create(const uvo_sharder_cfg* cfg, uvo_sharder** out) {
  if (!cfg || !out) return fail(UVO_E_BADARG, "null pointer");
  *out = nullptr;
  if (cfg->n_shards < 1 || cfg->n_shards > UVO_SHARD_MAX || cfg->chunk_frames < 1) return fail(UVO_E_BADARG, "bad sharder configuration");
  uvo_sharder* s = new uvo_sharder();
  s->cfg = *cfg;
  s->shards.resize(cfg->n_shards);
  int local = 0;
  for (int i = 0; i < cfg->n_shards; ++i) {
    Shard& sh = s->shards[i];
    sh.device = cfg->devices[i];
    if (sh.device == UVO_SHARD_REMOTE) continue;
    ++local;
    uvo_extractor_cfg ec = cfg->extractor;
    ec.device = sh.device;
    ec.max_batch = cfg->chunk_frames + 1;  // a chunk + its halo frame
    ec.max_input_keypoints = 0;
    int rc = uvo_extractor_create(&ec, &sh.ex);
    if (!rc) rc = uvo_extractor_set_pipeline(sh.ex, 2);
    if (rc) {
      uvo_sharder_destroy(s);
      return rc;
    }
    const int dcap = uvo_extractor_max_keypoints(sh.ex);
    s->dcap = dcap;
    if (cfg->match) {
      if (dcap > 65535) {
        uvo_sharder_destroy(s);
        return fail(UVO_E_UNSUPPORTED, "more than 65535 keypoints per frame: the all-pairs matcher packs train indices in 16 bits");
      }
      uvo_matcher_cfg mc;
      memset(&mc, 0, sizeof(mc));
      mc.max_query = dcap, mc.max_train = dcap, mc.max_batch = cfg->chunk_frames, mc.max_map_points = 0, mc.device = sh.device;
      rc = uvo_matcher_create(&mc, &sh.mt);
      if (rc) {
        uvo_sharder_destroy(s);
        return rc;
      }
      const size_t rows = (size_t)cfg->chunk_frames * dcap;
      bool ok = true;
      for (int l = 0; l < 2 && ok; ++l) {
        ok = hipMalloc((void**)&sh.d_idx0[l], rows * 4) == hipSuccess && hipMalloc((void**)&sh.d_idx1[l], rows * 4) == hipSuccess &&
             hipMalloc((void**)&sh.d_d0[l], rows * 2) == hipSuccess && hipMalloc((void**)&sh.d_d1[l], rows * 2) == hipSuccess &&
             hipEventCreateWithFlags(&sh.kernels_done[l], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&sh.rows_sent[l], hipEventDisableTiming) == hipSuccess;
      }
      if (!ok) {
        uvo_sharder_destroy(s);
        return fail(UVO_E_NOMEM, "sharder staging allocation failed");
      }
    }
  }
  if (local == 0) {
    uvo_sharder_destroy(s);
    return fail(UVO_E_BADARG, "no local shard (every device is UVO_SHARD_REMOTE)");
  }
  for (int i = 0; i < cfg->n_shards; ++i)
    if (s->shards[i].device != UVO_SHARD_REMOTE) s->shards[i].worker = std::thread(shard_worker, s, i);
  *out = s;
  return UVO_OK;
}

int uvo_sharder_max_keypoints(const uvo_sharder* s) { return s ? s->dcap : fail(UVO_E_BADARG, "null handle"); }

}  // extern "C"

namespace {

// A chunk in flight on one of the shard's two lanes.
struct InFlight {
  int ticket = -1;
  std::shared_ptr<Job> job;
  bool last_of_job = false, match = false;
};

void job_delivered(uvo_sharder* s, const std::shared_ptr<Job>& job, int rc, const char* msg) {
  std::lock_guard<std::mutex> lk(s->mu);
  if (rc != UVO_OK && job->rc == UVO_OK) job->rc = rc, job->err = msg;
  if (--job->remaining == 0) s->cv_done.notify_all();
}

// One shard's thread: its block of every job, chunk by chunk through the two lanes.
void shard_worker(uvo_sharder* s, int shard_index) {
  Shard& sh = s->shards[shard_index];
  (void)hipSetDevice(sh.device);
  (void)uvo_host_bind_near_device(sh.device, &sh.numa_node);  // the shard's staging and waiting happen next to its GPU's PCIe link
  const int C = s->cfg.chunk_frames, dcap = s->dcap;
  hipStream_t ms = sh.mt ? uvo_matcher_stream_internal(sh.mt) : nullptr;
  std::deque<InFlight> inflight;           // oldest first, at most two
  std::shared_ptr<Job> cur;                // the job whose chunks are being submitted
  int cur_f0 = 0, cur_end = 0;             // next chunk's first frame / end of the shard's block
  int last_seen = 0;                       // id of the last job taken from the queue
  bool cur_failed = false;
  char msg[600];
  auto retire_oldest = [&]() {
    InFlight f = inflight.front();
    inflight.pop_front();
    int rc = uvo_extract_batch_wait(sh.ex, f.ticket);  // (frees the lane also when it fails)
    // the matcher's rows are waited for whatever the extraction reported: the lane's buffers are about to be reused
    if (f.match && hipEventSynchronize(sh.rows_sent[f.ticket]) != hipSuccess && rc == UVO_OK) rc = fail(UVO_E_HIP, "hipEventSynchronize failed");
    if (rc != UVO_OK) {
      snprintf(msg, sizeof(msg), "shard %d: %s", shard_index, uvo_last_error());
      std::lock_guard<std::mutex> lk(s->mu);
      if (f.job->rc == UVO_OK) f.job->rc = rc, f.job->err = msg;
    }
    if (f.last_of_job) job_delivered(s, f.job, UVO_OK, "");
  };
  for (;;) {
    if (!cur) {  // take the next job, or -- with nothing queued -- finish what is in flight, or sleep
      std::unique_lock<std::mutex> lk(s->mu);
      std::shared_ptr<Job> nxt;
      for (const auto& j : s->jobs)
        if (j->id > last_seen) {
          nxt = j;
          break;
        }
      if (!nxt) {
        if (!inflight.empty()) {
          // nothing to submit right now: retire the oldest chunk once it has landed, but keep an eye on the queue meanwhile -- the
          // caller typically submits its next job the moment this one is delivered, and that job must not wait behind the chunk
          // that is still running on the other lane
          lk.unlock();
          const InFlight& f = inflight.front();
          const int done = uvo_extract_batch_done_internal(sh.ex, f.ticket);
          const bool rows = !f.match || hipEventQuery(sh.rows_sent[f.ticket]) != hipErrorNotReady;
          if (done != 0 && rows) {
            retire_oldest();
          } else {
            lk.lock();
            s->cv_work.wait_for(lk, std::chrono::microseconds(20));
          }
          continue;
        }
        if (s->stopping) return;
        s->cv_work.wait(lk);
        continue;
      }
      lk.unlock();
      last_seen = nxt->id;
      uvo_shard_plan pl;
      if (uvo_shard_plan_make(nxt->a.total, s->cfg.n_shards, shard_index, C, &pl) != UVO_OK || pl.n_frames == 0) {
        job_delivered(s, nxt, UVO_OK, "");  // nothing of this job lives here
        continue;
      }
      cur = nxt, cur_f0 = pl.first_frame, cur_end = pl.first_frame + pl.n_frames, cur_failed = false;
    }
    const RunArgs& a = cur->a;
    const bool match = s->cfg.match != 0 && a.idx0 != nullptr;
    const int f0 = cur_f0;
    const int nb = std::min(C, cur_end - f0);               // frames this chunk owns
    const int ne = nb + (f0 + nb < a.total ? 1 : 0);        // + the halo frame (the next chunk's / the neighbouring shard's first frame)
    const bool last = f0 + nb >= cur_end;
    if (inflight.size() == 2) retire_oldest();              // the lane about to be reused must have delivered its results
    int rc = UVO_OK, t = -1;
    const uint8_t* d_desc = nullptr;
    const int32_t* d_n = nullptr;
    if (!cur_failed) {
      // the lane this chunk will run on: its `kernels_done` event is recorded right behind the extraction kernels, in front of the lane's
      // result copies, so that the matcher starts under the download instead of behind it
      const int ln = uvo_extractor_next_lane_internal(sh.ex);
      rc = uvo_extract_batch_submit_internal(sh.ex, ne, nb, a.imgs + (ptrdiff_t)(f0 - a.imgs_first_frame) * a.frame_stride, a.width, a.height, a.stride,
                                             a.frame_stride, a.out_kp + (size_t)f0 * a.cap, a.out_desc + (size_t)f0 * a.cap * 32, a.cap, a.n_out + f0, &t,
                                             match ? sh.kernels_done[ln] : nullptr, &d_desc, &d_n);
      if (rc == UVO_OK && match) {
        const int np = ne - 1;  // pairs (f0 + j, f0 + j + 1)
        // the matcher reads the lane's descriptors in HBM: its stream waits for the lane's kernels (not for the lane's downloads);
        // the lane's next batch waits for the matcher through retire_oldest()
        if (hipStreamWaitEvent(ms, sh.kernels_done[t], 0) != hipSuccess) rc = fail(UVO_E_HIP, "hipStreamWaitEvent failed");
        if (rc == UVO_OK && np > 0) {
          rc = uvo_hamming_knn2_batch_device(sh.mt, np, d_desc, d_n, dcap, d_desc + (size_t)dcap * 32, d_n + 1, dcap, sh.d_idx0[t], sh.d_d0[t], sh.d_idx1[t],
                                             sh.d_d1[t]);
          const size_t row4 = (size_t)dcap * 4, row2 = (size_t)dcap * 2;
          // (pitched copies on purpose, also when the pitches agree: they overlap the other lane's upload, large linear ones do not)
          if (rc == UVO_OK &&
              (hipMemcpy2DAsync(a.idx0 + (size_t)f0 * a.cap, (size_t)a.cap * 4, sh.d_idx0[t], row4, row4, np, hipMemcpyDeviceToHost, ms) != hipSuccess ||
               hipMemcpy2DAsync(a.idx1 + (size_t)f0 * a.cap, (size_t)a.cap * 4, sh.d_idx1[t], row4, row4, np, hipMemcpyDeviceToHost, ms) != hipSuccess ||
               hipMemcpy2DAsync(a.d0 + (size_t)f0 * a.cap, (size_t)a.cap * 2, sh.d_d0[t], row2, row2, np, hipMemcpyDeviceToHost, ms) != hipSuccess ||
               hipMemcpy2DAsync(a.d1 + (size_t)f0 * a.cap, (size_t)a.cap * 2, sh.d_d1[t], row2, row2, np, hipMemcpyDeviceToHost, ms) != hipSuccess))
            rc = fail(UVO_E_HIP, "copy of the knn-2 rows failed");
        }
        if (rc == UVO_OK && hipEventRecord(sh.rows_sent[t], ms) != hipSuccess) rc = fail(UVO_E_HIP, "hipEventRecord failed");
        // the matcher part failed half way: whatever did get enqueued (the knn-2 kernel reading this lane's descriptors, copies into the
        // caller's arrays) must have landed before the error is reported -- the caller may free those arrays, the lane is resubmitted
        if (rc != UVO_OK) (void)hipStreamSynchronize(ms);
      }
      if (rc != UVO_OK) {
        snprintf(msg, sizeof(msg), "shard %d: %s", shard_index, uvo_last_error());
        std::lock_guard<std::mutex> lk(s->mu);
        if (cur->rc == UVO_OK) cur->rc = rc, cur->err = msg;
        cur_failed = true;
      }
    }
    if (t >= 0) {
      InFlight f;
      f.ticket = t, f.job = cur, f.last_of_job = last, f.match = match && !cur_failed;
      inflight.push_back(f);
    } else if (last) {  // (error path) nothing in flight carries the job's end: let what is in flight land first
      while (!inflight.empty()) retire_oldest();
      job_delivered(s, cur, UVO_OK, "");
    }
    cur_f0 += nb;
    if (last) cur.reset();
  }
}

}  // namespace

extern "C" {

int uvo_sharder_submit(uvo_sharder* s, const uint8_t* imgs, int n_imgs, int imgs_first_frame, int total_frames, int width, int height,
                       ptrdiff_t stride, ptrdiff_t frame_stride, uvo_keypoint* out_kp, uint8_t* out_desc, int cap, int32_t* n_out, int32_t* idx0, uint16_t* d0,
                       int32_t* idx1, uint16_t* d1, int* ticket) {
  if (!s || !imgs || !out_kp || !out_desc || !n_out || !ticket) return fail(UVO_E_BADARG, "null pointer");
  *ticket = 0;
  if (total_frames < 1 || imgs_first_frame < 0 || n_imgs < 1 || width < 1 || height < 1 || stride < width || frame_stride < (ptrdiff_t)stride * (height - 1) + width)
    return fail(UVO_E_BADARG, "bad frame count / geometry");
  if (width > s->cfg.extractor.max_width || height > s->cfg.extractor.max_height) return fail(UVO_E_BADARG, "image size outside what the sharder was sized for");
  if (cap < s->dcap) return fail(UVO_E_CAPACITY, "cap must be at least uvo_sharder_max_keypoints()");
  const bool any_match_ptr = idx0 || d0 || idx1 || d1, all_match_ptr = idx0 && d0 && idx1 && d1;
  if (any_match_ptr && (!all_match_ptr || !s->cfg.match)) return fail(UVO_E_BADARG, "match outputs need all four arrays and a sharder created with match = 1");
  // the local shards' frames (+ halo) must lie inside the n_imgs frames `imgs` holds from imgs_first_frame on
  int local = 0;
  for (int i = 0; i < s->cfg.n_shards; ++i) {
    if (s->shards[i].device == UVO_SHARD_REMOTE) continue;
    ++local;
    uvo_shard_plan pl;
    int rc = uvo_shard_plan_make(total_frames, s->cfg.n_shards, i, s->cfg.chunk_frames, &pl);
    if (rc) return rc;
    if (pl.n_frames > 0 && pl.first_frame < imgs_first_frame) return fail(UVO_E_BADARG, "imgs does not hold a local shard's first frame");
    if (pl.n_frames > 0 && pl.first_frame + pl.n_frames + (pl.halo_frame >= 0 ? 1 : 0) > imgs_first_frame + n_imgs)
      return fail(UVO_E_BADARG, "imgs ends before a local shard's last frame / halo frame");
  }
  auto job = std::make_shared<Job>();
  job->a = RunArgs{imgs, imgs_first_frame, total_frames, width, height, stride, frame_stride, out_kp, out_desc, cap, n_out, idx0, idx1, d0, d1};
  job->remaining = local;
  {
    std::lock_guard<std::mutex> lk(s->mu);
    job->id = s->next_id++;
    s->jobs.push_back(job);
    *ticket = job->id;
  }
  s->cv_work.notify_all();
  return UVO_OK;
}

int uvo_sharder_wait(uvo_sharder* s, int ticket) {
  if (!s) return fail(UVO_E_BADARG, "null handle");
  std::unique_lock<std::mutex> lk(s->mu);
  std::shared_ptr<Job> job;
  for (const auto& j : s->jobs)
    if (j->id == ticket) job = j;
  if (!job) return fail(UVO_E_BADARG, "no such job (already waited for?)");
  s->cv_done.wait(lk, [&] { return job->remaining == 0; });
  for (auto it = s->jobs.begin(); it != s->jobs.end(); ++it)
    if ((*it)->id == ticket) {
      s->jobs.erase(it);
      break;
    }
  lk.unlock();
  if (job->rc != UVO_OK) return fail(job->rc, job->err.c_str());
  return UVO_OK;
}

int uvo_sharder_run(uvo_sharder* s, const uint8_t* imgs, int n_imgs, int imgs_first_frame, int total_frames, int width, int height,
                    ptrdiff_t stride, ptrdiff_t frame_stride, uvo_keypoint* out_kp, uint8_t* out_desc, int cap, int32_t* n_out, int32_t* idx0, uint16_t* d0,
                    int32_t* idx1, uint16_t* d1) {
  int ticket = 0;
  int rc = uvo_sharder_submit(s, imgs, n_imgs, imgs_first_frame, total_frames, width, height, stride, frame_stride, out_kp, out_desc, cap, n_out, idx0, d0, idx1,
                              d1, &ticket);
  if (rc) return rc;
  return uvo_sharder_wait(s, ticket);
}

}  // extern "C"
