#!/bin/bash
# build libepihip_t<tag>.so from a variant of cx_report.hip (or mhl_report.hip with FILE=mhl_report): scratch/build_variant.sh <variant.hip> <tag> [extra flags]
set -e
SRC=$(realpath $1); TAG=$2; shift 2
FILE=${FILE:-cx_report}
cd /root/repo/epialleler_amd/csrc
make -s libepihip.so
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fno-fast-math -ffp-contract=off -I. -I../../include "$@" -x hip -c $SRC -o /tmp/${FILE}_t$TAG.o
OBJS=$(ls engine.o util.o tiles.o per_read.o cx_report.o mhl_report.o mhl_fused.o match_target.o patterns.o synth.o comm.o layout.o capi.o bam_pack.o report_writer.o host_common.o | grep -v "^${FILE}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libepihip_t$TAG.so $OBJS /tmp/${FILE}_t$TAG.o -lz -ldl -lpthread -Wl,-rpath,/opt/rocm/lib
ls -la libepihip_t$TAG.so
