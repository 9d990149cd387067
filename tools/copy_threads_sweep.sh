#!/bin/bash
# staging-fill thread count vs end-to-end rate (one engine), 4 000 x 1 MiB
for t in 2 4 6 8 10 12; do
  echo -n "SNAPHASH_COPY_THREADS=$t  "
  SNAPHASH_COPY_THREADS=$t timeout -k 10 200 python tools/e2e_engines.py 4000 2>/dev/null | grep "engines on GPU 0: 1"
done
