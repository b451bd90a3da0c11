/* fpv_create on whatever device the (preloaded, faked) runtime describes: prints one line of JSON with what the handle decided -
 * the cache model's verdict, the automatic rotation for n drones, the row stride - for tests/test_device_guard.py to hold against
 * the rule.  usage: driver <n>   (the default drone parameters are irrelevant here: no kernel runs) */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/fpv_abi.h"

int main(int argc, char** argv)
{
    const int64_t n = argc > 1 ? atoll(argv[1]) : 1 << 20;
    fpv_params_t p;
    FILE* f = fopen(argc > 2 ? argv[2] : "", "rb");                  /* a packed fpv_params_t written by the test (fpyv_amd._lib.pack_params) */
    if (!f || fread(&p, sizeof p, 1, f) != 1) { fprintf(stderr, "cannot read the params blob\n"); return 2; }
    fclose(f);
    fpv_handle_t h = 0;
    int rc = fpv_create(&p, n, 0, &h);
    if (rc != FPV_OK) { fprintf(stderr, "fpv_create: %s\n", fpv_last_error()); return 3; }
    fpv_cache_model_t m, d;
    if (fpv_get_cache_model(h, &m) != FPV_OK || fpv_device_cache_model(0, &d) != FPV_OK || memcmp(&m, &d, sizeof m)) return 4;
    int64_t rot = -1, rot_explicit = -1;
    if (fpv_get_rotation(h, &rot) != FPV_OK) return 5;
    char why[512];
    snprintf(why, sizeof why, "%s", fpv_last_error());
    if (fpv_set_rotation(h, 4096) != FPV_OK || fpv_get_rotation(h, &rot_explicit) != FPV_OK) return 6;
    printf("{\"matches\": %d, \"arch\": \"%s\", \"compute_units\": %d, \"xcds\": %d, \"l2\": %lld, \"mall\": %lld, \"rotation\": %lld, "
           "\"rotation_explicit_4096\": %lld, \"ld_device\": %lld, \"ld_model\": %lld, \"reason\": \"%s\", \"last_error_after_get_rotation\": \"%s\"}\n",
           m.matches, m.arch, m.compute_units, m.xcds, (long long)m.l2_bytes_per_xcd, (long long)m.infinity_cache_bytes, (long long)rot,
           (long long)rot_explicit, (long long)fpv_recommended_ld_device(n, 0), (long long)fpv_recommended_ld(n), m.reason, why);
    fpv_destroy(h);
    return 0;
}
