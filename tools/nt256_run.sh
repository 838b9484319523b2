#!/bin/bash
# runs the prototype variants (tools/_bin/nt256_*) on the shapes of the configs[4] cell launches; each under a timeout
cd $GRAFT_REPO_ROOT
for sh in "62208 768 768" "62208 768 1536" "96000 768 1536" "8192 8192 8192"; do
  for v in ${NT256_VARIANTS:-base nostore samerow noprio}; do
    echo "== $v $sh"; timeout 60 tools/_bin/nt256_$v $sh 10 2>&1 | tail -4
  done
done
