// tools/fuzz/oracle_asan.c -- the checker (oracle/) decoding and optimizing the files on the command line; built with
// AddressSanitizer / UBSan by tests/test_sanitizers_cpu.py.  File names go to stderr so that a report names its input.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "jpegref.h"
int main(int argc, char **argv) {
    for (int a = 1; a < argc; a++) {
        FILE *f = fopen(argv[a], "rb");
        if (!f) continue;
        fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
        unsigned char *d = malloc(n); fread(d, 1, n, f); fclose(f);
        fprintf(stderr, "%s\n", argv[a]);
        jref_decoder *dec = jref_create();
        jref_info info; memset(&info, 0, sizeof info);
        jref_set_input(dec, d, n);
        int rc = jref_identify(dec, 0, &info);
        jref_destroy(dec);
        if (rc == 0 && info.width > 0 && info.height > 0 && info.ncomp > 0) {
            size_t cap = (size_t)info.width * info.height * info.ncomp;
            unsigned char *out = calloc(cap, 1);
            char err[256];
            jref_decode_to_8bit(d, n, info.ncomp, out, cap, &info, err, 256);
            free(out);
        }
        { uint8_t *o = NULL; size_t ol = 0; char err[256]; if (jref_optimize(d, n, 1, &o, &ol, err, 256) == 0) free(o); }
        free(d);
    }
    return 0;
}
