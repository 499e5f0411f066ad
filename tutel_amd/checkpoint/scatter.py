"""CLI of tutel.checkpoint.scatter (reference: tutel/checkpoint/scatter.py:10-17, same flags)."""
import argparse

from .reshard import scatter


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--output_size", type=int, required=True)
    parser.add_argument("--input", type=str, required=True)
    parser.add_argument("--outputs", type=str, required=True)
    parser.add_argument("--namespace", type=str, default="")
    args = parser.parse_args()
    for f in scatter(args.input, args.outputs, args.output_size, args.namespace):
        print(f"Model params have been scattered to: {f}")


if __name__ == "__main__":
    main()
