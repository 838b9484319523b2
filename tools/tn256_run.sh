#!/bin/bash
# runs the weight-gradient prototype variants (tools/_bin/tn256_*) on the shapes of the configs[4] cell launches + ragged checks; each under a timeout
cd $GRAFT_REPO_ROOT
for sh in "5000 304 520 3" "777 256 256 2" "62208 768 768 0" "62208 768 1536 0" "62208 1536 1536 0" "62208 2304 1536 0" "62208 768 768 4"; do
  for v in ${TN256_VARIANTS:-base nostore}; do
    echo "== $v $sh"; timeout 120 tools/_bin/tn256_$v $sh 10 2>&1 | tail -6
  done
done
