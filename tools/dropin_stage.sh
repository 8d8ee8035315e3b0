#!/bin/bash
# Build container only: stage a scratch copy of the reference's Python packages for ONE gpurun call (tools/dropin_gpu.sh).
# _refcopy/ is git-ignored and must be removed right after the call (tools/dropin_stage.sh --clean): reference sources never
# enter the repository.
ROOT=$(cd $(dirname $0)/.. && pwd)
if [ "${1:-}" = "--clean" ]; then rm -rf $ROOT/_refcopy; echo removed; exit 0; fi
mkdir -p $ROOT/_refcopy/torchlie $ROOT/_refcopy/torchkin
cp -r /root/reference/theseus $ROOT/_refcopy/theseus
cp -r /root/reference/torchlie/torchlie $ROOT/_refcopy/torchlie/torchlie
cp -r /root/reference/torchkin/torchkin $ROOT/_refcopy/torchkin/torchkin
cp -r /root/reference/examples $ROOT/_refcopy/examples
find $ROOT/_refcopy -name "*.so" -delete; find $ROOT/_refcopy -name "__pycache__" -type d -exec rm -rf {} +
du -sh $ROOT/_refcopy
