/* Harness for tests/test_wal_framing.py::test_scan_under_sanitizers: runs rgb_wal_scan (count mode, then
 * store mode) over files given on the command line, each loaded into an exactly-sized heap block so that
 * AddressSanitizer reports any read past the end.  Built with g++ -fsanitize=address,undefined together
 * with ra_amd/csrc/rgb_wal_host.cpp (no HIP needed). */
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include "ra_gpu_wal.h"
int main(int argc, char **argv) {
  for (int a = 1; a < argc; ++a) {
    FILE *f = fopen(argv[a], "rb"); if (!f) return 2;
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    unsigned char *buf = (unsigned char *)malloc(n ? n : 1);      /* exactly sized: ASan sees any overrun */
    if (fread(buf, 1, n, f) != (size_t)n) return 3;
    fclose(f);
    uint32_t cnt = 0, n2 = 0, end = 0; uint64_t consumed = 0;
    int rc = rgb_wal_scan(buf, n, NULL, 0, &cnt, &consumed, &end);
    if (rc == 0) {
      rgb_wal_scanned *recs = (rgb_wal_scanned *)malloc(sizeof(rgb_wal_scanned) * (cnt ? cnt : 1));
      rc = rgb_wal_scan(buf, n, recs, cnt, &n2, &consumed, &end);
      if (rc || n2 != cnt) return 4;
      free(recs);
    }
    printf("%d %u %llu %u\n", rc, cnt, (unsigned long long)consumed, end);
    free(buf);
  }
  return 0;
}
