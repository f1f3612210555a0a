#!/bin/bash
# gpurun wrapper: leaves the commit hash where the GPU box (which gets no .git) can read it, then forwards to gpurun.
#   tools/gpu.sh [--timeout S] -- '<command>'
cd "$(dirname "$0")/.." && git rev-parse --short HEAD > .git_rev 2>/dev/null
exec /usr/local/graft/bin/gpurun "$@"
