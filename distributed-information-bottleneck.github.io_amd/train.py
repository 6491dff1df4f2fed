"""`python -m dib_amd.train` - the reference's `train.py` Keras path on the MI355X-native engine.

Mirrors reference train.py:12-178 (same flag names and defaults; the argparse `type=bool` trap and the
`infonce_space_dimensionality` typo of the reference are fixed): dataset dict ->
DistributedIBNet -> compile -> callbacks -> fit -> History post-processing (beta / KL in bits / loss without
the KL term) -> distributed information plane PNG.  The InfoNCE custom loop (train.py:180-289) is a
"next" row and is rejected explicitly.

Multi-GPU: launch with `python -m torch.distributed.run --nproc-per-node N -m dib_amd.train ...`; every rank
runs the same script, gradients are all-reduced over RCCL.
"""
from __future__ import annotations

import argparse
import os

import numpy as np


def _bool(v):
    if isinstance(v, bool):
        return v
    return str(v).lower() in ("1", "true", "yes", "y", "t")


def get_args(argv=None):
    p = argparse.ArgumentParser(description="Distributed IB training (MI355X-native)")
    p.add_argument('--dataset', default='boolean_circuit', choices=['boolean_circuit', 'synthetic_tabular', 'double_pendulum'])
    p.add_argument('--data_path', type=str, default='./data/')
    p.add_argument('--artifact_outdir', type=str, default='./training_artifacts/')
    p.add_argument('--ib', type=_bool, default=False, help='vanilla IB: treat all features as one (train.py:111-113)')
    p.add_argument('--learning_rate', type=float, default=3e-4)
    p.add_argument('--beta_start', type=float, default=1e-4)
    p.add_argument('--beta_end', type=float, default=3e0)
    p.add_argument('--number_pretraining_epochs', type=int, default=10 ** 3)
    p.add_argument('--number_annealing_epochs', type=int, default=10 ** 4)
    p.add_argument('--batch_size', type=int, default=128)
    p.add_argument('--use_positional_encoding', type=_bool, default=True)
    p.add_argument('--activation_fn', type=str, default='relu')
    p.add_argument('--feature_embedding_dimension', type=int, default=32)
    p.add_argument('--optimizer', type=str, default='adam')
    p.add_argument('--save_compression_matrices_frequency', type=int, default=0)
    p.add_argument('--feature_encoder_architecture', type=int, nargs='+', default=[128, 128])
    p.add_argument('--number_positional_encoding_frequencies', type=int, default=5)
    p.add_argument('--integration_network_architecture', type=int, nargs='+', default=[256, 256])
    p.add_argument('--infonce_loss', type=_bool, default=False)
    p.add_argument('--infonce_shared_dimensionality', type=int, default=64)
    p.add_argument('--infonce_y_encoder_architecture', type=int, nargs='+', default=[128, 128])
    p.add_argument('--infonce_similarity', type=str, default='l2')
    p.add_argument('--infonce_temperature', type=float, default=1.)
    p.add_argument('--pendulum_time_delta', type=float, default=2)
    p.add_argument('--pendulum_number_trajectories', type=int, default=0, help='0 = simulator default (1000)')
    p.add_argument('--boolean_random_circuit', type=_bool, default=False)
    p.add_argument('--boolean_number_input_gates', type=int, default=10)
    p.add_argument('--synthetic_rows', type=int, default=1 << 20)
    p.add_argument('--synthetic_features', type=int, default=64)
    p.add_argument('--seed', type=int, default=0, help='noise / init / shuffle seed (the reference is unseeded)')
    p.add_argument('--verbose', type=_bool, default=False)
    return p.parse_args(argv)


def postprocess_history(h, number_features, loss_is_info_based):
    """reference train.py:168-178: per-epoch series from `history.history` - the task loss without its beta * sum KL term, KL
    in bits, info-based losses in bits (same dtypes: float32 beta / loss series updated in place, float64 KL) - plus the
    validation series the reference forgot to build."""
    F = number_features
    beta_series = np.float32(h['beta'])
    kl_series = np.stack([h[f'KL{f}'] for f in range(F)], -1)
    loss_series = np.float32(h['loss'])
    loss_series_validation = np.float32(h['val_loss'])
    loss_series -= beta_series * np.sum(kl_series, axis=-1)
    kl_series /= np.log(2)
    kl_series_validation = np.stack([h[f'val_KL{f}'] for f in range(F)], -1)
    loss_series_validation -= np.float32(h['val_beta']) * np.sum(kl_series_validation, axis=-1)
    kl_series_validation /= np.log(2)
    if loss_is_info_based:
        loss_series /= np.log(2)
        loss_series_validation /= np.log(2)
    return dict(beta=beta_series, kl_bits=kl_series, loss=loss_series, kl_bits_validation=kl_series_validation,
                loss_validation=loss_series_validation)


def main(argv=None):
    from . import data, models, optimizers, visualization
    args = get_args(argv)
    import torch
    import torch.distributed as dist
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group("nccl")
    rank = dist.get_rank() if dist.is_initialized() else 0
    number_epochs = args.number_pretraining_epochs + args.number_annealing_epochs
    os.makedirs(args.artifact_outdir, exist_ok=True)
    dataset_dict = data.DATASETS[args.dataset](
        data_path=args.data_path, boolean_random_circuit=args.boolean_random_circuit,
        boolean_number_input_gates=args.boolean_number_input_gates, synthetic_rows=args.synthetic_rows,
        synthetic_features=args.synthetic_features, pendulum_time_delta=args.pendulum_time_delta,
        pendulum_number_trajectories=args.pendulum_number_trajectories, seed=args.seed)
    if dataset_dict['loss'] == 'infonce' and not args.infonce_loss:
        raise ValueError(f"dataset {args.dataset} trains with --infonce_loss True (reference data.py:131)")
    if rank == 0:
        print(f'Dataset {args.dataset} loaded.')
    if args.ib:  # train.py:111-113
        dataset_dict['feature_dimensionalities'] = [int(np.sum(dataset_dict['feature_dimensionalities']))]
        dataset_dict['number_features'] = 1
    activation = None if args.activation_fn in ('None', 'none', '') else args.activation_fn
    model = models.DistributedIBNet(
        dataset_dict['feature_dimensionalities'], args.feature_encoder_architecture,
        args.integration_network_architecture,
        dataset_dict['output_dimensionality'] if not args.infonce_loss else args.infonce_shared_dimensionality,  # train.py:116
        use_positional_encoding=args.use_positional_encoding,
        number_positional_encoding_frequencies=args.number_positional_encoding_frequencies, activation_fn=activation,
        feature_embedding_dimension=args.feature_embedding_dimension,
        output_activation_fn=dataset_dict['output_activation_fn'] if not args.infonce_loss else None,
        noise_seed=args.seed, init_seed=args.seed, shuffle_seed=args.seed)
    if args.infonce_loss:  # ---- custom training loop, reference train.py:180-289 ----
        from . import infonce
        F = dataset_dict['number_features']
        out = infonce.fit_infonce(
            model, dataset_dict['x_train'], dataset_dict['y_train'], dataset_dict['x_valid'], dataset_dict['y_valid'],
            batch_size=args.batch_size, number_pretraining_epochs=args.number_pretraining_epochs,
            number_annealing_epochs=args.number_annealing_epochs, beta_start=args.beta_start, beta_end=args.beta_end,
            learning_rate=args.learning_rate, y_encoder_architecture=args.infonce_y_encoder_architecture,
            shared_dimensionality=args.infonce_shared_dimensionality, similarity=args.infonce_similarity,
            temperature=args.infonce_temperature, use_positional_encoding=args.use_positional_encoding,
            number_positional_encoding_frequencies=args.number_positional_encoding_frequencies, activation_fn=activation,
            seed=args.seed)
        # train.py:271-286: KL and (info based) losses to bits
        kl_series, kl_series_validation = out['kl'] / np.log(2), out['kl_validation'] / np.log(2)
        loss_series, loss_series_validation = out['loss_infonce'], out['loss_infonce_validation']
        if dataset_dict['loss_is_info_based']:
            loss_series, loss_series_validation = loss_series / np.log(2), loss_series_validation / np.log(2)
        if rank == 0:
            print('Finished training.')
            np.savez(os.path.join(args.artifact_outdir, 'history.npz'), beta=np.float32(out['beta']), kl_bits=kl_series,
                     loss=loss_series, kl_bits_validation=kl_series_validation, loss_validation=loss_series_validation)
            visualization.save_distributed_info_plane(kl_series_validation, loss_series_validation, args.artifact_outdir)
        return out
    optimizer = optimizers.get(args.optimizer)
    optimizer.learning_rate = args.learning_rate  # train.py:128-129
    model.compile(optimizer=optimizer, loss=dataset_dict['loss'], metrics=dataset_dict['metrics'])
    callbacks = [models.InfoBottleneckAnnealingCallback(args.beta_start, args.beta_end, args.number_pretraining_epochs,
                                                        args.number_annealing_epochs)]
    if args.save_compression_matrices_frequency > 0:
        callbacks.append(models.SaveCompressionMatricesCallback(
            args.save_compression_matrices_frequency, dataset_dict['x_valid'],
            dataset_dict.get('x_valid_raw', dataset_dict['x_valid']), args.artifact_outdir))
    history = model.fit(dataset_dict['x_train'], dataset_dict['y_train'], epochs=number_epochs, shuffle=True,
                        batch_size=args.batch_size, callbacks=callbacks, verbose=args.verbose,
                        validation_data=(dataset_dict['x_valid'], dataset_dict['y_valid']))
    series = postprocess_history(history.history, dataset_dict['number_features'], dataset_dict['loss_is_info_based'])
    beta_series, kl_series, loss_series = series['beta'], series['kl_bits'], series['loss']
    kl_series_validation, loss_series_validation = series['kl_bits_validation'], series['loss_validation']
    if rank == 0:
        print('Finished training.')
        np.savez(os.path.join(args.artifact_outdir, 'history.npz'), beta=beta_series, kl_bits=kl_series,
                 loss=loss_series, kl_bits_validation=kl_series_validation, loss_validation=loss_series_validation)
        visualization.save_distributed_info_plane(kl_series_validation, loss_series_validation, args.artifact_outdir)
    return history


if __name__ == '__main__':
    main()
