"""Command-line options: drop-in for the sampling subset of reference `utils/parser_util.py`.

Same flag names, defaults and groups (`:53-58` base, `:61-67` diffusion, `:70-95` model,
`:99-106` dataset, `:141-154` sampling, `:157-171` generate) and the same rule that the
dataset / model / diffusion groups are overwritten from `args.json` next to the checkpoint
(`:7-33`).  The reference's quirks are kept on purpose: `type=bool` flags are truthy for any
non-empty string (`:55,67`), `--dataset`'s default is outside its own choices (`:101`).
Additive flags live in the 'native' group and never collide with reference names.
"""
import json
import os
from argparse import ArgumentParser


CHECKPOINT_GROUPS = ("dataset", "model", "diffusion")     # option groups a checkpoint's args.json decides (:13)


def parse_and_load_from_model(parser, argv=None):
    """Parse the command line, then let the `args.json` stored next to the checkpoint decide every option of the
    dataset / model / diffusion groups (reference `utils/parser_util.py:7-33`): a model is sampled with the settings it
    was trained with, whatever the command line says.  Options the file does not know keep their value, with the
    reference's warning.  A model trained without condition dropout cannot be guided, so its guidance scale becomes 1."""
    for add in (add_data_options, add_model_options, add_diffusion_options, add_native_options):
        add(parser)
    args = parser.parse_args(argv)
    if not (args.synthetic and not args.model_path):     # additive: --synthetic without a checkpoint has no args.json
        stored_path = os.path.join(os.path.dirname(args.model_path), "args.json")
        assert os.path.exists(stored_path), "Arguments json file was not found!"
        with open(stored_path) as f:
            stored = json.load(f)
        for name in (n for title in CHECKPOINT_GROUPS for n in get_args_per_group_name(parser, args, title)):
            if name in stored:
                setattr(args, name, stored[name])
            else:
                print("Warning: was not able to load [{}], using default value [{}] instead.".format(name, getattr(args, name)))
    if args.cond_mask_prob == 0:
        args.guidance_param = 1
    return args


def get_args_per_group_name(parser, args, group_name):
    """Destination names of the options declared in the argparse group titled `group_name`.  (The reference *returns* a
    ValueError instance for an unknown title instead of raising it, `:36-41`; iterating that fails with TypeError -- kept.)"""
    for group in parser._action_groups:
        if group.title == group_name:
            return [action.dest for action in group._group_actions]
    return ValueError("group_name was not found.")


def add_base_options(parser):
    group = parser.add_argument_group('base')
    group.add_argument("--cuda", default=True, type=bool, help="Use cuda device, otherwise use CPU.")
    group.add_argument("--device", default=0, type=int, help="Device id to use.")
    group.add_argument("--seed", default=10, type=int, help="For fixing random seed.")
    group.add_argument("--batch_size", default=256, type=int, help="Batch size during training.")


def add_diffusion_options(parser):
    group = parser.add_argument_group('diffusion')
    group.add_argument("--noise_schedule", default='cosine', choices=['linear', 'cosine'], type=str)
    group.add_argument("--diffusion_steps", default=1000, type=int,
                       help="Parsed but ignored, exactly like the reference (utils/model_util.py:40).")
    group.add_argument("--sigma_small", default=True, type=bool, help="Use smaller sigma values.")


def add_model_options(parser):
    group = parser.add_argument_group('model')
    group.add_argument("--arch", default='trans_enc', choices=['trans_enc', 'trans_dec', 'gru'], type=str)
    group.add_argument("--emb_trans_dec", default=False, type=bool)
    group.add_argument("--layers", default=8, type=int, help="Number of layers.")
    group.add_argument("--latent_dim", default=256, type=int, help="Transformer width.")
    group.add_argument("--cond_mask_prob", default=.1, type=float)
    group.add_argument("--lambda_rcxyz", default=0.0, type=float)
    group.add_argument("--lambda_vel", default=0.0, type=float)
    group.add_argument("--lambda_fc", default=0.0, type=float)
    group.add_argument("--unconstrained", action='store_true')
    group.add_argument("--use_text", action='store_true')
    group.add_argument("--use_audio", action='store_true')
    group.add_argument("--mfcc_input", action='store_true')
    group.add_argument("--use_wav_enc", action='store_true')
    group.add_argument("--seed_poses", type=int, default=10)


def add_data_options(parser):
    group = parser.add_argument_group('dataset')
    group.add_argument("--dataset", default='humanml', choices=['genea2022', 'genea2023'], type=str)
    group.add_argument("--data_dir", default="", type=str)
    group.add_argument("--num_frames", default=120, type=int)


def add_sampling_options(parser):
    group = parser.add_argument_group('sampling')
    group.add_argument("--model_path", default='', type=str,
                       help="Path to model####.pt (required unless --synthetic).")
    group.add_argument("--output_dir", default='', type=str)
    group.add_argument("--num_samples", default=10, type=int)
    group.add_argument("--num_repetitions", default=3, type=int)
    group.add_argument("--guidance_param", default=2.5, type=float)


def add_generate_options(parser):
    group = parser.add_argument_group('generate')
    group.add_argument("--motion_length", default=6.0, type=float)
    group.add_argument("--input_text", default='', type=str)
    group.add_argument("--action_file", default='', type=str)
    group.add_argument("--text_prompt", default='', type=str)
    group.add_argument("--action_name", default='', type=str)


def add_native_options(parser):
    """Additive flags of the MI355X build (none exist in the reference)."""
    group = parser.add_argument_group('native')
    group.add_argument("--synthetic", action='store_true',
                       help="Random weights + N(0,1) seed poses / MFCCs (no dataset, no checkpoint needed).")
    group.add_argument("--synthetic_njoints", default=0, type=int,
                       help="Pose channels for --synthetic when --dataset is not a GENEA set (e.g. 263).")
    group.add_argument("--arch_version", default='mdm', choices=['mdm', 'mdm_old'],
                       help="mdm = V2 (model/mdm.py), mdm_old = V1 encoder-only topology (model/mdm_old.py).")
    group.add_argument("--sampler", default='p', choices=['p', 'ddim'],
                       help="p_sample_loop (reference default) or ddim_sample_loop.")
    group.add_argument("--timestep_respacing", default='', type=str, help="e.g. ddim100; '' = all 1000 steps.")
    group.add_argument("--eta", default=0.0, type=float)
    group.add_argument("--rng", default=None, choices=['torch', 'philox'],
                       help="torch = the reference's generator and draw order (single-GPU default); philox = in-kernel "
                            "counter-based noise keyed by the global sample index (shard invariant; multi-GPU default).")
    group.add_argument("--progress", action='store_true', help="tqdm bar over the denoising steps (reference: always on).")
    group.add_argument("--chunks", default=14, type=int, help="Autoregressive chunks per take (reference: 14).")
    group.add_argument("--synthetic_audio", action='store_true',
                       help="With --synthetic: derive y['mfcc'] from synthetic audio through the GPU MFCC front end "
                            "(dataset.py:81-95) instead of drawing N(0,1) features.")
    group.add_argument("--packed_cache", default='', type=str,
                       help="Directory of packed-weight images: the first run of a checkpoint writes the kernels' operand "
                            "layout there, later runs upload it instead of unpickling and repacking the checkpoint.")
    group.add_argument("--compute_dtype", default='fp32', choices=['fp32', 'fp16', 'bf16'],
                       help="fp32 = exact fp32 MFMA (reference precision); fp16 / bf16 = 16-bit MFMA operands and activation "
                            "stream, fp32 accumulate (bf16 trades 3 significant bits for fp32's exponent range).")


def generate_args(argv=None):
    parser = ArgumentParser()
    add_base_options(parser)
    add_sampling_options(parser)
    add_generate_options(parser)
    return parse_and_load_from_model(parser, argv)
