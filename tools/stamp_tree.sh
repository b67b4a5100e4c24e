#!/bin/bash
# Writes TREE_ID (git-ignored, travels with the gpurun snapshot): the commit the GPU box's copy corresponds to, "-dirty" when
# the working tree differs from it.  Run before a gpurun call whose outputs are to be committed under profiles/.
cd "$(dirname "$0")/.."
id=$(git rev-parse --short HEAD)
git diff --quiet HEAD -- . ':!TREE_ID' || id="$id-dirty"
echo "$id" > TREE_ID
cat TREE_ID
