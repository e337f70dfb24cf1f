#!/usr/bin/env python
"""`python train_teacher.py --teacher SAGE --dataset ...`: the reference's teacher entry point
(reference train_teacher.py) on the MI355X hot path; flags, YAML semantics and artefacts in glnn_amd/cli.py."""
from glnn_amd.cli import teacher_main

if __name__ == "__main__":
    teacher_main()
