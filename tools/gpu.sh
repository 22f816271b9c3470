#!/bin/bash
# tools/gpu.sh [--timeout S] -- '<command>': gpurun with the commit (and whether the tree is dirty) left in .fl_commit, so that
# profiles written on the GPU box (which has no .git) can say which code they measured.
cd "$(dirname "$0")/.."
c=$(git rev-parse --short HEAD 2>/dev/null)
if [ -n "$(git status --porcelain --untracked-files=no 2>/dev/null)" ]; then c="$c+dirty"; fi
echo "$c" > .fl_commit
exec /usr/local/graft/bin/gpurun "$@"
