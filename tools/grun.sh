#!/bin/bash
# gpurun wrapper: stamps the tree with the commit it was taken from (the GPU box has no .git) and forwards to gpurun.
#   tools/grun.sh [--timeout S] -- '<command>'
cd "$(dirname "$0")/.." || exit 1
c=$(git rev-parse --short HEAD 2>/dev/null || echo unknown)
git diff --quiet HEAD 2>/dev/null || c="${c}+dirty"
echo "$c" > .combo_commit
exec /usr/local/graft/bin/gpurun "$@"
