#!/bin/bash
# gpurun wrapper: stamps the tree with the commit it was taken from (the GPU box has no .git), forwards to gpurun and retries
# while no GPU slot / box is free (exit code 3: nothing charged).
#   tools/grun.sh [--timeout S] -- '<command>'
cd "$(dirname "$0")/.." || exit 1
c=$(git rev-parse --short HEAD 2>/dev/null || echo unknown)
git diff --quiet HEAD 2>/dev/null || c="${c}+dirty"
echo "$c" > .combo_commit
for attempt in 1 2 3 4 5 6 7 8 9 10 11 12; do
  /usr/local/graft/bin/gpurun "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  echo "[grun] no slot (attempt $attempt), retrying in 60 s"
  sleep 60
done
exit 3
