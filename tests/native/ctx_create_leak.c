/* Leak check of the C ABI's failure paths (tests/test_sanitized_host.py builds and runs this under
 * AddressSanitizer with leak detection on): a ta_ctx_create that fails part-way must free what it took. */
#include <stdio.h>
#include "tissue_scan.h"

int main(void) {
    int failures = 0, created = 0;
    for (int round = 0; round < 50; ++round) {
        for (int dev = -1; dev < 3; ++dev) {
            ta_ctx* c = NULL;
            int rc = ta_ctx_create(dev, &c);
            if (rc == TA_OK) { ++created; if (ta_ctx_destroy(c) != TA_OK) return 2; }
            else { ++failures; if (c != NULL) return 3; }
        }
    }
    if (ta_ctx_destroy(NULL) != TA_OK) return 4;
    if (ta_extract(NULL, 31u, 10u) != TA_EINVAL) return 5;
    printf("ctx_create: %d failed cleanly, %d created and destroyed\n", failures, created);
    return 0;
}
