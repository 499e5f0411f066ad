"""CLI of tutel.checkpoint.gather (reference: tutel/checkpoint/gather.py:11-18, same flags)."""
import argparse

from .reshard import gather


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--input_size", type=int, required=True)
    parser.add_argument("--inputs", type=str, required=True)
    parser.add_argument("--output", type=str, required=True)
    parser.add_argument("--namespace", type=str, default="")
    parser.add_argument("--default_num_global_experts", type=int, default=0)
    args = parser.parse_args()
    gather(args.inputs, args.input_size, args.output, args.namespace, args.default_num_global_experts)
    print(f"Model params have been collected to: {args.output}")


if __name__ == "__main__":
    main()
