#!/usr/bin/env python3
"""Keypoint R-CNN training entry point with the flags of the reference's train_keypoints.py (:73-89): 17 COCO
keypoints, 56x56 heat maps, softmax cross entropy over positions (:21-27), n_fg_class = 1.  See train.py."""
from train import build_parser, run


def main():
    args = build_parser(keypoints=True).parse_args()
    run(args, keypoints=True)


if __name__ == '__main__':
    main()
