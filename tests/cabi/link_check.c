/*
 * The C ABI from plain C: this file includes include/vq_hip.h as C99, links against libvqhip.so
 * and calls the host-only entry points (version, status strings, size helpers, the argument checks
 * that return before any HIP call).  No GPU is needed.  Built and run by tests/test_cabi.py.
 */
#include <stdio.h>
#include <string.h>

#include "vq_hip.h"

#define CHECK(cond)                                             \
  do {                                                          \
    if (!(cond)) {                                              \
      fprintf(stderr, "link_check: failed: %s (line %d)\n", #cond, __LINE__); \
      return 1;                                                 \
    }                                                           \
  } while (0)

int main(void) {
  CHECK(vq_version() >= 100);
  CHECK(strcmp(vq_status_string(VQ_OK), vq_status_string(VQ_ERR_INVALID_ARG)) != 0);
  /* sizes at the benchmark shape */
  CHECK(vq_image_bytes(8192, 256) == (size_t)(8192 / 64) * (64 * 256 + 256) * 4);
  CHECK(vq_norms_bytes(8192) == 8192 * 4);
  CHECK(vq_search_blocks(4194304) == 16384);
  CHECK(vq_loss_partial_bytes(4194304) >= (size_t)16384 * 4);
  CHECK(vq_search_uses_persistent_kernel(4194304, 8192, 256, VQ_LAYOUT_NCHW, 1024) == 1);
  CHECK(vq_search_uses_persistent_kernel(4096, 512, 64, VQ_LAYOUT_NCHW, 256) == 0);
  CHECK(vq_control_bytes() >= 16 * 4);
  CHECK(vq_filter_work_bytes(4194304) == (size_t)4194304 * 168);
  CHECK(vq_filter_image_bytes(8192, 256) > (size_t)8192 * 256 * 2);
  CHECK(vq_scatter_slabs_for(4194304, 8192, 256, VQ_LAYOUT_NCHW, 1024) >= 1);
  int parts[2] = {0, 0};
  CHECK(vq_filter_split(65536, 8192, parts) == VQ_OK && parts[0] == 2 && parts[1] == 32);
  /* argument checks return before anything touches the device */
  CHECK(vq_search(NULL, VQ_LAYOUT_NCHW, 16, 64, 16, NULL, NULL, NULL, 8, 1.0f, 0, NULL, NULL, NULL, NULL, NULL, NULL,
                  NULL, NULL, NULL) == VQ_ERR_INVALID_ARG);
  CHECK(vq_gather_project(NULL, VQ_LAYOUT_ND, 4, 8, 1, NULL, 4, NULL, NULL, 8, NULL, NULL, NULL) == VQ_ERR_INVALID_ARG);
  CHECK(vq_backward_gather(NULL, NULL, NULL, VQ_LAYOUT_ND, 4, 8, 1, NULL, 4, NULL, 1.0f, NULL, NULL) == VQ_ERR_INVALID_ARG);
  CHECK(vq_filter_split(-1, 8192, parts) == VQ_ERR_INVALID_ARG);
  printf("link_check ok (libvqhip version %d)\n", vq_version());
  return 0;
}
