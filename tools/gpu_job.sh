#!/bin/bash
bash tools/profile_round.sh r04_n
