#!/bin/bash
# Builds tools/host_sanitize.cpp with the library's host-only sources under AddressSanitizer and
# UndefinedBehaviorSanitizer (g++, no HIP) and runs it.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $ROOT/build/san
g++ -std=c++17 -O1 -g -fno-omit-frame-pointer -fsanitize=address,undefined -fno-sanitize-recover=undefined -I$ROOT/include \
    $ROOT/tools/host_sanitize.cpp $ROOT/strugatzki_amd/csrc/sgz_audiofile.cpp $ROOT/strugatzki_amd/csrc/sgz_config.cpp \
    $ROOT/strugatzki_amd/csrc/sgz_select.cpp -o $ROOT/build/san/host_sanitize
D=$(mktemp -d)
$ROOT/build/san/host_sanitize $D
rm -rf $D
# the read-ahead threads of the database loops under ThreadSanitizer
g++ -std=c++17 -O1 -g -fno-omit-frame-pointer -fsanitize=thread -pthread -I$ROOT/include \
    $ROOT/tools/readahead_sanitize.cpp $ROOT/strugatzki_amd/csrc/sgz_audiofile.cpp -o $ROOT/build/san/readahead_sanitize
D=$(mktemp -d)
$ROOT/build/san/readahead_sanitize $D
rm -rf $D
