// Error plumbing shared by every entry point of libvnqa_hip.so.
#include "vnqa_common.h"
#include <string.h>
#include <stdlib.h>

static thread_local char g_err[512] = "";

void vnqa_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* vnqa_last_error(void) { return g_err; }
extern "C" int vnqa_version(void) { return VNQA_ABI_VERSION; }

// ---- CU partition between the full-chip stem kernels and a latency-bound chain of small kernels on another stream ----------
// A stream created by vnqa_stream_create_reserved(n) never runs on n of the chip's CUs (4 per XCD for n = 32): kernels on other
// streams (MACNetwork's ~1 000 dependent small launches per step, an RCCL all-reduce) find those CUs free at once instead of
// waiting for a stem workgroup that owns a whole CU's LDS / registers to retire.  The persistent one-workgroup-per-CU conv kernels
// size their grids to match through a PER-CALL descriptor field (VNQA_CONV_RESERVE_CUS in vnqa_conv_desc.flags): the library keeps
// no mutable global state and reads no environment variable.

extern "C" int vnqa_stream_create_reserved(int32_t reserve_cus, void** stream) {
  int dev = 0, n_cu = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) {
    vnqa_set_error("stream_create_reserved: cannot query the device");
    return VNQA_ERR_HIP;
  }
  VNQA_CHECK_ARG(n_cu % 64 == 0 && n_cu <= 512, "stream_create_reserved: unexpected CU count %d", n_cu);
  const int cls = n_cu / 8;                                    // CUs per residue class below
  VNQA_CHECK_ARG(stream != nullptr && reserve_cus >= 0 && reserve_cus % cls == 0 && reserve_cus <= n_cu - cls,
                 "stream_create_reserved: reserve_cus must be a multiple of %d in [0, %d] (got %d)", cls, n_cu - cls, reserve_cus);
  // Measured on MI355X (tools/probe_cu_mask.py, the whole frozen stem on the masked stream): mask bit i is CU i / 8 of XCD i % 8,
  // and inside an XCD consecutive indices fall on different shader engines.  Workgroups are dealt round-robin to XCDs and to
  // the engines of an XCD regardless of the mask, so the slowest XCD / engine sets the kernel's time: one missing CU in ONE XCD
  // costs 35 %, 4 missing CUs of the same engine in every XCD cost 85 % — while bits [0, 32 k) (k CUs off every engine of every
  // XCD) cost exactly their share (32 CUs: 6.6 -> 7.2 ms).  Hence only multiples of n_cu / 8 and only that set.
  uint32_t mask[16];
  for (int w = 0; w < 16; ++w) mask[w] = 0;
  for (int i = reserve_cus; i < n_cu; ++i) mask[i / 32] |= 1u << (i % 32);
  hipStream_t st = nullptr;
  if (hipExtStreamCreateWithCUMask(&st, (uint32_t)(n_cu / 32), mask) != hipSuccess) {
    vnqa_set_error("stream_create_reserved: hipExtStreamCreateWithCUMask failed");
    return VNQA_ERR_HIP;
  }
  *stream = (void*)st;
  return VNQA_OK;
}

// explicit CU mask (bit i of mask[i / 32] set = the stream may use CU i in the runtime's enumeration)
extern "C" int vnqa_stream_create_masked(const uint32_t* host_mask, int32_t words, void** stream) {
  VNQA_CHECK_ARG(host_mask != nullptr && words > 0 && stream != nullptr, "stream_create_masked: bad arguments");
  hipStream_t st = nullptr;
  if (hipExtStreamCreateWithCUMask(&st, (uint32_t)words, host_mask) != hipSuccess) {
    vnqa_set_error("stream_create_masked: hipExtStreamCreateWithCUMask failed");
    return VNQA_ERR_HIP;
  }
  *stream = (void*)st;
  return VNQA_OK;
}
