/* Host-side entry points of libfpv_hip.so that need no device, driven under AddressSanitizer + UBSan
 * (tests/test_sanitizers.py builds the library with `-Xarch_host -fsanitize=address,undefined` and this file against it):
 * the row-stride rule and its L2 set model over a few thousand populations, argument validation, error text. */
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/fpv_abi.h"

int main(void)
{
    int64_t checked = 0;
    uint64_t x = 88172645463325252ull;
    for (int i = 0; i < 3000; ++i) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        int64_t n = 1 + (int64_t)(x % ((uint64_t)1 << (10 + i % 19)));          /* 1 .. 2^28 */
        const int64_t ld = fpv_recommended_ld(n);
        if (ld < n || ld % 4 || ld > n + 4096) { printf("bad ld %lld for n %lld\n", (long long)ld, (long long)n); return 1; }
        ++checked;
    }
    const int64_t edges[] = {1, 63, 64, 65, 262144, 262145, 524288, 1048576, 1572864, 2097152, 2097153, (int64_t)1 << 28};
    for (unsigned i = 0; i < sizeof edges / sizeof edges[0]; ++i)
        if (fpv_recommended_ld(edges[i]) < edges[i]) return 2;
    if (fpv_recommended_ld(0) >= 0 || fpv_recommended_ld(-7) >= 0) return 3;
    if (fpv_abi_version() != FPV_ABI_VERSION) return 4;
    if (fpv_state_rows(0) != 14 || fpv_state_rows(99) != -1) return 5;
    if (fpv_sizeof(0) != (int)sizeof(fpv_params_t) || fpv_sizeof(1) != (int)sizeof(fpv_buffers_t) || fpv_sizeof(9) >= 0) return 6;
    if (!fpv_error_name(FPV_EINVAL) || !fpv_last_error()) return 7;
    fpv_params_t p;
    memset(&p, 0, sizeof p);
    p.struct_size = (uint32_t)sizeof p;
    fpv_handle_t h = 0;
    int rc = fpv_create(0, 16, 0, &h);
    if (rc >= 0 || h) return 8;                                                     /* null params */
    rc = fpv_create(&p, 0, 0, &h);
    if (rc >= 0 || h) return 9;                                                     /* n = 0 */
    rc = fpv_create(&p, ((int64_t)1 << 28) + 1, 0, &h);
    if (rc >= 0 || h) return 10;                                                    /* beyond the drone limit */
    /* (no create that gets as far as the device query: ROCm's ASan runtime cannot live in a process that initialises a GPU, and
     * this driver must also pass on a box that has one) */
    int64_t rot = -2;
    if (fpv_get_rotation(0, &rot) >= 0) return 12;                                  /* null handle */
    if (fpv_set_rotation(0, 128) >= 0) return 13;
    /* the cache-model rule (ABI 8) is host arithmetic: every text field is bounded, whatever the device calls itself */
    fpv_cache_model_t m;
    char longname[600];
    memset(longname, 'x', sizeof longname - 1);
    longname[sizeof longname - 1] = 0;
    memcpy(longname, "gfx950:", 7);
    if (fpv_check_cache_model(longname, 256, 4 << 20, &m) != FPV_OK || !m.matches || strlen(m.arch) != sizeof m.arch - 1) return 14;
    memcpy(longname, "gfx9999", 7);
    if (fpv_check_cache_model(longname, 256, 4 << 20, &m) != FPV_OK || m.matches || strlen(m.reason) >= sizeof m.reason || !strstr(m.reason, "is not gfx950")) return 15;
    if (fpv_check_cache_model("gfx950:sramecc+:xnack-", 32, 4 << 20, &m) != FPV_OK || m.matches || !strstr(m.reason, "32 compute units")) return 16;
    if (fpv_check_cache_model("gfx950", 256, 0, &m) != FPV_OK || !m.matches || m.xcds != 8 || m.reason[0]) return 17;
    if (fpv_check_cache_model(0, 256, 0, &m) >= 0 || fpv_check_cache_model("gfx950", 256, 0, 0) >= 0) return 18;
    if (fpv_sizeof(4) != (int)sizeof(fpv_cache_model_t) || m.struct_size != sizeof m) return 19;
    if (fpv_get_cache_model(0, &m) >= 0) return 20;
    if (!fpv_encoding_id(0) || !fpv_encoding_id(1) || fpv_encoding_id(2) || fpv_encoding_id(-1)) return 21;
    if (fpv_diag_xcd_map(0, 8, 0) >= 0) return 22;                                  /* refused before any device is touched */
    fpv_destroy(0);                                                                 /* destroying nothing is a no-op, never a crash */
    printf("host sanitizers: clean (%lld strides checked; last error text: %s)\n", (long long)checked, fpv_last_error());
    return 0;
}
