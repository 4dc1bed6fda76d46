/* Using libembodied_hip.so from plain C: the host index core needs no GPU.
 *
 *   gcc -std=c99 -Iinclude examples/c_abi_demo.c -Lembodied_amd -lembodied_hip -o demo
 *   LD_LIBRARY_PATH=embodied_amd:/opt/rocm/lib ./demo
 *
 * Prints the first Uniform(seed=0) draws over keys 0..9 -- the stream SURVEY.md
 * Appendix C captured from the reference: 8 6 5 2 3 0 0 0 1 8 6 9 5 6 9 7 --
 * and walks a small Replay index: 30 steps x 3 workers, length 5, capacity 50.
 */
#include <stdint.h>
#include <stdio.h>

#include "embodied_hip.h"

#define CHECK(call)                                                        \
  do {                                                                     \
    int32_t status_ = (call);                                              \
    if (status_ != EMB_OK) {                                               \
      fprintf(stderr, "%s -> %d: %s\n", #call, status_, emb_last_error()); \
      return 1;                                                            \
    }                                                                      \
  } while (0)

int main(void) {
  emb_selector_t* uniform = NULL;
  CHECK(emb_selector_create_uniform(0, &uniform));
  for (int64_t key = 0; key < 10; ++key) CHECK(emb_selector_insert(uniform, key, NULL, 0));
  printf("uniform:");
  for (int i = 0; i < 16; ++i) {
    int64_t key = -1;
    CHECK(emb_selector_sample(uniform, &key));
    printf(" %lld", (long long)key);
  }
  printf("\n");
  CHECK(emb_selector_destroy(uniform));

  emb_replay_config_t cfg = {0};
  cfg.length = 5;
  cfg.capacity = 50;
  cfg.chunksize = 8;
  cfg.n_slots = 64;
  cfg.owners = 1;
  emb_replay_t* replay = NULL;
  CHECK(emb_replay_create(&cfg, NULL, 0, &replay));
  for (int t = 0; t < 30; ++t) {
    for (int64_t worker = 0; worker < 3; ++worker) {
      int32_t row = -1;
      uint8_t stepid[EMB_STEPID_BYTES];
      CHECK(emb_replay_add_index(replay, 1, &worker, &row, stepid, NULL));
    }
  }
  int64_t items = 0;
  CHECK(emb_replay_len(replay, &items));
  int32_t rows[4 * 5];
  int64_t workers[4];
  CHECK(emb_replay_sample_index(replay, 4, EMB_MODE_TRAIN, rows, NULL, workers));
  printf("replay: %lld items; sampled workers", (long long)items);
  for (int b = 0; b < 4; ++b) printf(" %lld", (long long)workers[b]);
  printf("\n");
  CHECK(emb_replay_destroy(replay));
  return 0;
}
