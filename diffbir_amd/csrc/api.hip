// Error plumbing + ABI version for libdbir_hip.so (see include/dbir.h).
#include <stdarg.h>

#include "common.h"

static thread_local char g_err[512] = "";

void dbir_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* dbir_last_error(void) { return g_err; }
extern "C" int dbir_abi_version(void) { return 5; }

void dbir_attention_set_variant(int v);  // attention.hip
void dbir_xf_set_variant(int v);         // xformer.hip

extern "C" int dbir_set_option(int key, int value) {
  switch (key) {
    case DBIR_OPT_ATTN_VARIANT:
      DBIR_CHECK_ARG((value >= 2 && value <= 8) || (value >= 1000 && value <= 1000 + (1 << 20)),
                     "dbir_set_option: attention variant must be 2 (default), 3 (no LDS-resident cross kernel), 4 / 5 / 6 (4-wave kernel with the "
                     "pre-round-4 softmax at 4 / 3 / default waves per SIMD), 8 (the 8-wave attn3 experiment for long self-attentions), or 1000 + n "
                     "(query-row threshold of attn3)");
      dbir_attention_set_variant(value);
      return DBIR_OK;
    case DBIR_OPT_XF_VARIANT:
      DBIR_CHECK_ARG(value >= 0 && value <= 7, "dbir_set_option: fused-transformer staging variant must be 0 .. 7");
      dbir_xf_set_variant(value);
      return DBIR_OK;
  }
  dbir_set_error("dbir_set_option: unknown key %d", key);
  return DBIR_ERR_ARG;
}
