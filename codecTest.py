#!/usr/bin/env python3
"""Dataset synthesis with RTF on the HIP path -- same command line as the reference's codecTest.py:120-147.

    python codecTest.py --subset clean_test --encoder exp/.../checkpoint-200000steps.pkl \
        --decoder exp/.../checkpoint-500000steps.pkl --output_dir output

The data location comes from the encoder's config.yml (data.path / data.subset[<subset>]), as in the reference.
"""
import argparse

from audiodec_amd.offline import TestMain


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--subset", type=str, default="clean_test")
    parser.add_argument("--subset_num", type=int, default=-1)
    parser.add_argument("--encoder", type=str, required=True)
    parser.add_argument("--decoder", type=str, required=True)
    parser.add_argument("--output_dir", type=str, required=True)
    parser.add_argument("--specific_folder", choices=("True", "False"), default="False")
    args = parser.parse_args()

    test_main = TestMain(args=args)
    test_main.load_dataset(args.subset, args.subset_num)
    test_main.load_encoder()
    test_main.load_decoder()
    test_main.initial_folder(args.subset, args.output_dir, args.specific_folder)
    test_main.run()


if __name__ == "__main__":
    main()
