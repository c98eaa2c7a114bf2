#!/usr/bin/env python3
"""Code-vector statistics (mean / scale for the vocoder's input normalisation) on the HIP path -- same
command line as the reference's codecStatistic.py:116-138.

    python codecStatistic.py -c config/statistic/symAD_vctk_48000_hop300_clean.yaml --subset train
"""
import argparse

from audiodec_amd.offline import StatisticMain


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("-c", "--config", type=str, required=True)
    parser.add_argument("--subset", type=str, default="train")
    parser.add_argument("--subset_num", type=int, default=-1)
    args = parser.parse_args()

    statistic_main = StatisticMain(args=args)
    statistic_main.load_dataset(args.subset, args.subset_num)
    statistic_main.load_analyzer()
    statistic_main.run()


if __name__ == "__main__":
    main()
