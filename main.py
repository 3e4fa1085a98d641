#!/usr/bin/env python3
"""Inference CLI with the reference's interface (main.py:290-362):

    python main.py audio_file model_file [-o OUT.mid] [-d {cpu,cuda}] [-t THRESHOLD]

Exit code 1 with a message when a file is missing or transcription fails.  The checkpoint must be a
CNNRNNModelLarge(320, 512, 3) state_dict as in the reference (main.py:16-20); --model-type/--n-mels/
--hidden-size/--num-layers are additive overrides.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser(description="Transcribe audio files to MIDI using trained music transcription model")
    ap.add_argument("audio_file", type=str, help="Path to input audio file (wav)")
    ap.add_argument("model_file", type=str, help="Path to model checkpoint file (.pth)")
    ap.add_argument("-o", "--output", type=str, default=None, help="Path to output MIDI file (default: <audio_name>_transcription.mid)")
    ap.add_argument("-d", "--device", type=str, choices=["cpu", "cuda"], default=None, help="Device to use for inference (default: auto-detect)")
    ap.add_argument("-t", "--threshold", type=float, default=0.5, help="Threshold for note predictions (default: 0.5)")
    ap.add_argument("--model-type", default="cnn_rnn_large")
    ap.add_argument("--n-mels", type=int, default=320)
    ap.add_argument("--hidden-size", type=int, default=512)
    ap.add_argument("--num-layers", type=int, default=3)
    args = ap.parse_args()
    if not os.path.exists(args.audio_file):
        print(f"Error: Audio file not found: {args.audio_file}")
        sys.exit(1)
    if not os.path.exists(args.model_file):
        print(f"Error: Model file not found: {args.model_file}")
        sys.exit(1)
    print("=" * 60 + "\nMusic Transcription Pipeline\n" + "=" * 60)
    try:
        from music_transcription_amd.transcribe import transcribe_audio
        out = transcribe_audio(args.audio_file, args.model_file, args.output, args.device, args.threshold,
                               model_type=args.model_type, n_mels=args.n_mels, hidden_size=args.hidden_size,
                               num_layers=args.num_layers)
        print("=" * 60 + f"\nTranscription completed successfully!\nOutput: {out}\n" + "=" * 60)
    except Exception as e:
        print(f"Error during transcription: {e}")
        import traceback
        traceback.print_exc()
        sys.exit(1)


if __name__ == "__main__":
    main()
