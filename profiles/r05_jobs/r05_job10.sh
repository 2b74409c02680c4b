#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
bash tools/edits_exp.sh "11 10 11 10 0"
