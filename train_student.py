#!/usr/bin/env python
"""`python train_student.py --teacher SAGE --student MLP3w8 --dataset ...`: the reference's distillation
entry point (reference train_student.py) on the MI355X hot path; see glnn_amd/cli.py."""
from glnn_amd.cli import student_main

if __name__ == "__main__":
    student_main()
