#!/usr/bin/env python3
"""Drop-in for the reference's pose_estimation/export_predicted_poses_real.py (:126-236): same six
required arguments, same opencv_poses.json + overlay JPEGs; the per-frame cv2.solvePnPRansac
loop is one batched launch of the HIP EPnP+RANSAC kernel.  --no_overlay skips the debug JPEGs
(image I/O dominates the reference script's wall-clock; the poses do not depend on it)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import scpose  # noqa: E402,F401
from importlib import import_module  # noqa: E402

pose_export = import_module("spacecraft-pose-estimation_amd.pose_export")


def main():
    parser = argparse.ArgumentParser(description="event frames to pose estimation results.")
    parser.add_argument("--frames_dir", required=True, type=str, help="directory with blender event frames")
    parser.add_argument("--detection_annotations", required=True, type=str, help="file with object detection results in COCO format")
    parser.add_argument("--pose_annotations", required=True, type=str, help="file with pose estimation results")
    parser.add_argument("--landmarks_file", required=True, type=str, help="blender landmarks file")
    parser.add_argument("--calibration_file_path", required=True, type=str, help="file with camera calibration parameters")
    parser.add_argument("--output_dir", required=True, type=str, help="output directory")
    parser.add_argument("--no_overlay", action="store_true", help="do not write the per-frame reprojection JPEGs")
    parser.add_argument("--with_status", action="store_true", help="add the per-frame RANSAC status to the JSON records")
    args = parser.parse_args()
    pose_export.export(args.frames_dir, args.detection_annotations, args.pose_annotations, args.landmarks_file,
                       args.calibration_file_path, args.output_dir, overlay=not args.no_overlay,
                       include_status=args.with_status)


if __name__ == "__main__":
    main()
