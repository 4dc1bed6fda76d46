// Entry points that are one kernel launch each: obs stack, action mask, row
// gather / scatter by env id, windowing, the return scans, the synthetic env.
#include "handles.h"

extern "C" {

int32_t emb_device_count(int32_t* count) {
  return guarded([&] {
    need(count, "count is null");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    *count = (e == hipSuccess) ? n : 0;
  });
}

// A HIP stream whose kernels run on `n_cus` compute units only, beginning with
// unit `first_cu` of the driver's own numbering (which deals consecutive units
// round-robin over the XCDs and their shader engines, so a range is spread
// evenly over the chip).
int32_t emb_stream_create_on_cus(int32_t first_cu, int32_t n_cus, void** stream_out) {
  return guarded([&] {
    need(stream_out && first_cu >= 0 && n_cus >= 1, "stream_create_on_cus: bad arguments");
    int device = 0;
    HIP_OK(hipGetDevice(&device));
    hipDeviceProp_t prop;
    HIP_OK(hipGetDeviceProperties(&prop, device));
    const int total = prop.multiProcessorCount;
    need(first_cu + n_cus <= total, "stream_create_on_cus: the range exceeds the device's compute units");
    std::vector<uint32_t> mask(static_cast<size_t>((total + 31) / 32), 0u);
    for (int cu = first_cu; cu < first_cu + n_cus; ++cu) mask[cu / 32] |= 1u << (cu % 32);
    hipStream_t s = nullptr;
    HIP_OK(hipExtStreamCreateWithCUMask(&s, static_cast<uint32_t>(mask.size()), mask.data()));
    {
      std::lock_guard<std::mutex> lock(g_cu_streams.mu);
      g_cu_streams.streams.emplace_back(s, n_cus);
      g_cu_streams.count.store(static_cast<int>(g_cu_streams.streams.size()), std::memory_order_release);
    }
    *stream_out = s;
  });
}

int32_t emb_stream_destroy(void* stream) {
  return guarded([&] {
    hipStream_t s = static_cast<hipStream_t>(stream);
    need(s != nullptr, "stream_destroy: null stream");
    {
      std::lock_guard<std::mutex> lock(g_cu_streams.mu);
      auto& v = g_cu_streams.streams;
      v.erase(std::remove_if(v.begin(), v.end(), [&](const auto& e) { return e.first == s; }), v.end());
      g_cu_streams.count.store(static_cast<int>(v.size()), std::memory_order_release);
    }
    HIP_OK(hipStreamDestroy(s));
  });
}

// ------------------------------------------------------------------ kernels --

int32_t emb_obs_stack(const void* src, const int32_t* env_ids, int64_t n, int64_t pixels,
                      int64_t channels, int32_t layout, int32_t out_dtype, float scale,
                      float offset, void* dst, void* stream) {
  return guarded([&] {
    need(src && dst && n >= 0 && pixels > 0 && channels > 0, "obs_stack: bad arguments");
    need(layout == EMB_LAYOUT_SAME || layout == EMB_LAYOUT_CHANNELS_FIRST, "obs_stack: bad layout");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (!env_ids) {
      HIP_OK(emb::launch_obs_stack(static_cast<const uint8_t*>(src), nullptr, dst, n, pixels,
                                   channels, layout, out_dtype, scale, offset, s));
      return;
    }
    std::lock_guard<std::mutex> lock(g_ring_mu);
    auto lease = global_ring().acquire(n * sizeof(int32_t), s);
    std::memcpy(lease.host, env_ids, n * sizeof(int32_t));
    global_ring().upload(lease, n * sizeof(int32_t), s);
    HIP_OK(emb::launch_obs_stack(static_cast<const uint8_t*>(src),
                                 reinterpret_cast<const int32_t*>(lease.device), dst, n, pixels,
                                 channels, layout, out_dtype, scale, offset, s));
    global_ring().retire(lease, s);
  });
}

int32_t emb_copy_bytes(const void* src, void* dst, int64_t bytes, void* stream) {
  return guarded([&] {
    need(bytes >= 0 && (bytes == 0 || (src && dst)), "copy_bytes: bad arguments");
    HIP_OK(emb::launch_copy_bytes(src, dst, bytes, static_cast<hipStream_t>(stream)));
  });
}

int32_t emb_mask_actions(const void* act, void* out, int64_t n, int64_t row_elems, int32_t dtype,
                         const void* is_last, void* stream) {
  return guarded([&] {
    need(act && out && is_last && n >= 0 && row_elems >= 0, "mask_actions: bad arguments");
    HIP_OK(emb::launch_mask_rows(act, out, n, row_elems, dtype, static_cast<const uint8_t*>(is_last),
                                 static_cast<hipStream_t>(stream)));
  });
}

int32_t emb_mask_actions_notify(const void* act, void* out, int64_t n, int64_t row_elems, int32_t dtype,
                                const void* is_last, void* counter, void* flag, uint32_t seq, void* stream) {
  return guarded([&] {
    need(act && out && is_last && counter && flag && n > 0 && row_elems > 0, "mask_actions_notify: bad arguments");
    HIP_OK(emb::launch_mask_rows(act, out, n, row_elems, dtype, static_cast<const uint8_t*>(is_last),
                                 static_cast<hipStream_t>(stream), static_cast<uint32_t*>(counter),
                                 static_cast<uint32_t*>(flag), seq));
  });
}

static void rows_move(void* table, int64_t rowbytes, const int32_t* ids, int64_t n, void* batch,
                      bool gather, hipStream_t s) {
  need(table && batch && ids && rowbytes > 0 && n >= 0, "rows_gather/scatter: bad arguments");
  if (n == 0) return;
  std::lock_guard<std::mutex> lock(g_ring_mu);
  auto lease = global_ring().acquire(n * sizeof(int32_t), s);
  std::memcpy(lease.host, ids, n * sizeof(int32_t));
  global_ring().upload(lease, n * sizeof(int32_t), s);
  emb::MovePlan plan;
  plan.n_keys = 1;
  plan.key[0] = {static_cast<uint8_t*>(table), static_cast<uint8_t*>(batch), rowbytes};
  plan.n_rows = static_cast<int32_t>(n);
  plan.rows = reinterpret_cast<const int32_t*>(lease.device);
  HIP_OK(gather ? emb::launch_gather(plan, s) : emb::launch_scatter(plan, s));
  global_ring().retire(lease, s);
}

int32_t emb_rows_gather(const void* table, int64_t rowbytes, const int32_t* ids, int64_t n, void* dst,
                        void* stream) {
  return guarded([&] { rows_move(const_cast<void*>(table), rowbytes, ids, n, dst, true, static_cast<hipStream_t>(stream)); });
}

int32_t emb_rows_scatter(void* table, int64_t rowbytes, const int32_t* ids, int64_t n, const void* src,
                         void* stream) {
  return guarded([&] { rows_move(table, rowbytes, ids, n, const_cast<void*>(src), false, static_cast<hipStream_t>(stream)); });
}

int32_t emb_window(const void* src, void* dst, int64_t batch, int64_t total, int64_t start,
                   int64_t count, int64_t rowbytes, void* stream) {
  return guarded([&] {
    need(src && dst && batch >= 0 && start >= 0 && count >= 0 && start + count <= total && rowbytes > 0,
         "window: bad arguments");
    HIP_OK(emb::launch_window(static_cast<const uint8_t*>(src), static_cast<uint8_t*>(dst), batch,
                              total, start, count, rowbytes, static_cast<hipStream_t>(stream)));
  });
}

int32_t emb_window_keys(int32_t n_keys, const void* const* src, void* const* dst,
                        const int64_t* rowbytes, int64_t batch, int64_t total, int64_t start,
                        int64_t count, void* stream) {
  return guarded([&] {
    need(n_keys >= 1 && src && dst && rowbytes && batch >= 0 && start >= 0 && count >= 0 &&
         start + count <= total, "window_keys: bad arguments");
    if (batch == 0 || count == 0) return;
    need(batch * total <= INT32_MAX, "window_keys: batch too large");
    hipStream_t s = static_cast<hipStream_t>(stream);
    // A window is a gather whose "pool" is the source batch: sequence b is the
    // single span {b * total + start, count}.  All keys move in one launch.
    std::vector<int32_t> spans(3 * batch), rows;
    for (int64_t b = 0; b < batch; ++b) {
      spans[3 * b] = static_cast<int32_t>(b * total + start);
      spans[3 * b + 1] = static_cast<int32_t>(count);
      spans[3 * b + 2] = 0;
    }
    for (int lo = 0; lo < n_keys; lo += emb::kMaxKeys) {
      emb::MovePlan plan;
      for (int k = lo; k < std::min(n_keys, lo + emb::kMaxKeys); ++k) {
        need(src[k] && dst[k] && rowbytes[k] > 0, "window_keys: bad key");
        plan.key[plan.n_keys++] = {const_cast<uint8_t*>(static_cast<const uint8_t*>(src[k])),
                                   static_cast<uint8_t*>(dst[k]), rowbytes[k]};
      }
      plan.seq_len = static_cast<int32_t>(count);
      plan.n_rows = static_cast<int32_t>(batch * count);
      plan.spans_host = spans.data();
      plan.n_seq = static_cast<int32_t>(batch);
      if (emb::plan_fits_inline(plan)) {
        HIP_OK(emb::launch_gather(plan, s));
        continue;
      }
      if (rows.empty()) {
        rows.resize(batch * count);
        for (int64_t b = 0; b < batch; ++b)
          for (int64_t j = 0; j < count; ++j)
            rows[b * count + j] = static_cast<int32_t>(b * total + start + j);
      }
      plan.spans_host = nullptr;
      plan.n_seq = 0;
      std::lock_guard<std::mutex> lock(g_ring_mu);
      auto lease = global_ring().acquire(rows.size() * sizeof(int32_t), s);
      std::memcpy(lease.host, rows.data(), rows.size() * sizeof(int32_t));
      global_ring().upload(lease, rows.size() * sizeof(int32_t), s);
      plan.rows = reinterpret_cast<const int32_t*>(lease.device);
      HIP_OK(emb::launch_gather(plan, s));
      global_ring().retire(lease, s);
    }
  });
}

int32_t emb_scan_gae(const void* rew, const void* val, const void* last, const void* term, int64_t B,
                     int64_t T, float live_scale, float lam, void* adv, void* tar, void* stream) {
  return guarded([&] {
    need(rew && val && last && term && adv && tar && B >= 0 && T >= 1, "scan_gae: bad arguments");
    HostLap hp;
    HIP_OK(emb::launch_gae(static_cast<const float*>(rew), static_cast<const float*>(val),
                           static_cast<const uint8_t*>(last), static_cast<const uint8_t*>(term), B, T,
                           live_scale, lam, static_cast<float*>(adv), static_cast<float*>(tar),
                           static_cast<hipStream_t>(stream)));
    hp.lap(20, "gae: launch");
  });
}

int32_t emb_scan_gae_grouped(const void* rew, const void* val, const void* last, const void* term,
                             int64_t B, int64_t T, float live_scale, float lam, void* adv,
                             void* tar, int64_t group, int64_t group_stride, void* stream) {
  return guarded([&] {
    need(rew && val && last && term && adv && tar && B >= 0 && T >= 1 && group >= 0 &&
             group_stride >= 0 && group_stride % 4 == 0, "scan_gae_grouped: bad arguments");
    HIP_OK(emb::launch_gae(static_cast<const float*>(rew), static_cast<const float*>(val),
                           static_cast<const uint8_t*>(last), static_cast<const uint8_t*>(term), B, T,
                           live_scale, lam, static_cast<float*>(adv), static_cast<float*>(tar),
                           static_cast<hipStream_t>(stream), group, group_stride));
  });
}

int32_t emb_scan_lambda(const void* last, const void* term, const void* rew, const void* boot,
                        int64_t B, int64_t T, float disc, float lam, void* ret, void* stream) {
  return guarded([&] {
    need(last && term && rew && boot && ret && B >= 0 && T >= 1, "scan_lambda: bad arguments");
    HIP_OK(emb::launch_lambda_return(static_cast<const uint8_t*>(last), static_cast<const uint8_t*>(term),
                                     static_cast<const float*>(rew), static_cast<const float*>(boot), B, T,
                                     disc, lam, static_cast<float*>(ret), static_cast<hipStream_t>(stream)));
  });
}

int32_t emb_scan_lambda_multi(int32_t n_problems, const emb_lambda_problem_t* problems, void* stream) {
  return guarded([&] {
    need(n_problems >= 0 && (problems || n_problems == 0), "scan_lambda_multi: bad arguments");
    std::vector<emb::LambdaProblem> list;
    list.reserve(n_problems);
    for (int i = 0; i < n_problems; ++i) {
      const emb_lambda_problem_t& q = problems[i];
      need(q.B >= 0 && q.T >= 1, "scan_lambda_multi: bad shape");
      if (q.B == 0 || q.T < 2) continue;
      need(q.last && q.term && q.rew && q.boot && q.ret, "scan_lambda_multi: null buffer");
      list.push_back({static_cast<const uint8_t*>(q.last), static_cast<const uint8_t*>(q.term),
                      static_cast<const float*>(q.rew), static_cast<const float*>(q.boot),
                      static_cast<float*>(q.ret), q.B, q.T, q.disc, q.lam});
    }
    HostLap hp;
    HIP_OK(emb::launch_lambda_return_multi(static_cast<int>(list.size()), list.data(),
                                           static_cast<hipStream_t>(stream)));
    hp.lap(23, "lambda-return (multi): launch");
  });
}

int32_t emb_scan_director(const void* rew, const void* cont, const void* value, int64_t T, int64_t B,
                          float discount, float lam, void* ret, void* stream) {
  return guarded([&] {
    need(rew && cont && value && ret && B >= 0 && T >= 1, "scan_director: bad arguments");
    HIP_OK(emb::launch_director_score(static_cast<const float*>(rew), static_cast<const float*>(cont),
                                      static_cast<const float*>(value), T, B, discount, lam,
                                      static_cast<float*>(ret), static_cast<hipStream_t>(stream)));
  });
}

int32_t emb_abstract_traj(const void* reward, const void* cont, int64_t T, int64_t B, int32_t k,
                          void* reward_out, void* cont_out, void* stream) {
  return guarded([&] {
    need(cont && (reward || !reward_out) && T >= 1 && B >= 0 && k >= 1, "abstract_traj: bad arguments");
    HIP_OK(emb::launch_abstract_traj(static_cast<const float*>(reward), static_cast<const float*>(cont),
                                     T, B, k, static_cast<float*>(reward_out),
                                     static_cast<float*>(cont_out), static_cast<hipStream_t>(stream)));
  });
}

int32_t emb_synth_env_step(void* image, void* reward, void* is_first, void* is_last, void* is_terminal,
                           int64_t n, int64_t frame_bytes, int64_t env0, int64_t episode_len,
                           const void* reset, void* counters, int32_t turn, void* stream) {
  return guarded([&] {
    need(image && reward && is_first && is_last && is_terminal && counters && n >= 0 && episode_len >= 1,
         "synth_env_step: bad arguments");
    HostLap hp;
    HIP_OK(emb::launch_synth_env(static_cast<uint8_t*>(image), static_cast<float*>(reward),
                                 static_cast<uint8_t*>(is_first), static_cast<uint8_t*>(is_last),
                                 static_cast<uint8_t*>(is_terminal), n, frame_bytes, env0, episode_len,
                                 static_cast<const uint8_t*>(reset), static_cast<int32_t*>(counters),
                                 turn, static_cast<hipStream_t>(stream)));
    hp.lap(21, "synthetic env: launch");
  });
}

}  // extern "C"
