#!/bin/bash
# tools/variant.sh <name>: build the library and keep a copy as variants/<name>.so (kernel A/B runs on one GPU box: tools/ab_variants.sh)
set -e
make -s -C pime-robust-non-linear-set-point-control-with-reinforcement-learning_amd/csrc -j8
mkdir -p variants && cp pime-robust-non-linear-set-point-control-with-reinforcement-learning_amd/libpime_hip.so variants/$1.so && echo "variants/$1.so"
