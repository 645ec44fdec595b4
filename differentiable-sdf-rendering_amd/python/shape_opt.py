"""`optimize_shape` (python/shape_opt.py:32-132): the SDF reconstruction loop -- Adam, per-iteration
view batch (one batched primal launch + one gradient-pass launch instead of a Python loop over
views), regulariser, checkpoints, gradient scrubbing, redistancing, parameter averaging."""
import os
import time
from os.path import join

import numpy as np
import torch

import dsdf
from dsdf import parallel
from integrators.reparam import Scene, create_integrator, render, traverse
from shapes import Grid3d, create_sphere_sdf
from util import dump_metadata, read_image, resize_img, set_sensor_res, write_image
from variables import Adam


def load_ref_images(paths, multiscale=False, device='cuda'):
    """Reference images and their resolution pyramid (python/shape_opt.py:16-29): {width: image}."""
    if not multiscale:
        return [read_image(fn, device) for fn in paths]
    result = []
    for fn in paths:
        img = read_image(fn, device)
        d = {int(img.shape[1]): img}
        res = np.array(img.shape[:2])
        while res.min() > 4:
            res = res // 2
            d[int(res[1])] = resize_img(img, res, smooth=True)
        result.append(d)
    return result


def build_scene(scene_config, config, device='cuda'):
    """Stands in for mi.load_file(scene.xml, shape_file='dummysdf.xml', sdf_filename=..., integrator=...)."""
    init = create_sphere_sdf([16, 16, 16], device=device)
    props = {'sdf': Grid3d(init)}
    if any(k.endswith('roughness.volume.data') or k.endswith('base_color.volume.data') for k in scene_config.param_keys):
        props.update(base_color=0.5, roughness=0.5)                    # principled-* configs: the scene's BSDF is `principled`
    integ = create_integrator(config.integrator, props)
    return Scene(scene_config.sensors, integ)


def optimize_shape(scene_config, mts_args, ref_image_paths, output_dir, config, write_ldr_images=True):
    if mts_args:
        print(f"Cmdline arguments passed to Mitsuba: {mts_args} (ignored: no scene file)")
    ref_images = load_ref_images(ref_image_paths, True)
    sdf_scene = build_scene(scene_config, config)
    integ = sdf_scene.integrator()
    sdf_object = integ.sdf
    integ.warp_field = config.get_warpfield(sdf_object)
    params = traverse(sdf_scene)
    params.keep(scene_config.param_keys)
    missing = [k for k in scene_config.param_keys if k not in params]
    if missing:
        raise NotImplementedError(f"parameters {missing} are not published by integrator '{config.integrator}' "
                                  f"(reflectance / base_color / roughness volumes need sdf_direct_reparam)")
    opt = Adam(lr=config.learning_rate, params=params, mask_updates=config.mask_optimizer)
    n_iter = config.n_iter
    scene_config.initialize(opt, sdf_scene)
    params.update(opt)
    rank, world = (torch.distributed.get_rank(), torch.distributed.get_world_size()) \
        if torch.distributed.is_available() and torch.distributed.is_initialized() else (0, 1)

    with torch.no_grad():                                                  # shape initialisation renders
        imgs = render(sdf_scene, params, list(scene_config.sensors), seed=0, spp=config.spp * config.primal_spp_mult)
    if rank == 0:
        for idx in range(len(scene_config.sensors)):
            write_image(join(output_dir, f'init-{idx:02d}.npy'), imgs[idx])
    for sensor in scene_config.sensors:
        set_sensor_res(sensor, scene_config.init_res)

    opt_image_dir = join(output_dir, 'opt')
    os.makedirs(opt_image_dir, exist_ok=True)
    seed, loss_values, t_start = 0, [], time.time()
    n_sens = len(scene_config.sensors)
    try:
        for i in range(n_iter):
            batch = list(scene_config.get_sensor_iterator(i))
            mine = parallel.strided_view_shard(batch, rank, world)
            seeds_per_view = 1 + n_sens
            loss = torch.zeros((), device='cuda')
            if mine:
                idxs = [b[0] for b in mine]
                sens = [b[1] for b in mine]
                # the reference advances `seed` by 1 + n_sensors per rendered view (shape_opt.py:78-81)
                base = seed + rank * seeds_per_view
                imgs = render(sdf_scene, params, sens, seed=base, spp=config.spp * config.primal_spp_mult,
                              seed_grad=base + seeds_per_view, spp_grad=config.spp)
                width = sens[0].film_size()[0]
                for j, idx in enumerate(idxs):
                    loss = loss + scene_config.loss(imgs[j], ref_images[idx][width]) / scene_config.batch_size
                loss.backward()
                if rank == 0 and write_ldr_images:
                    for j, idx in enumerate(idxs):
                        write_image(join(opt_image_dir, f'opt-{i:04d}-{idx:02d}.png'),
                                    resize_img(imgs[j].detach(), scene_config.target_res))
            seed += seeds_per_view * len(batch)
            reg_loss = scene_config.eval_regularizer(opt, sdf_object, i)
            if isinstance(reg_loss, torch.Tensor) and reg_loss.requires_grad:
                (reg_loss / world).backward()
                loss = loss.detach() + reg_loss.detach() / world
            if world > 1:
                # gradients are reduced where autograd put them (a large tensor by its own in-place collective, small ones
                # through one persistent bucket); the scalar loss goes separately -- appending it to the list used to force a
                # copy of everything into a fresh flat bucket every iteration
                grads = [p.grad if p.grad is not None else torch.zeros_like(p) for _, p in opt.items()]
                loss = loss.detach().reshape(1).clone()
                work = parallel.all_reduce_gradients(grads, async_op=True)
                parallel.all_reduce_gradients([loss])
                if work is not None:
                    work.wait()
                loss = loss[0]
                for (_, p), g in zip(opt.items(), grads):
                    p.grad = g
            if rank == 0:
                scene_config.save_params(opt, output_dir, i, force=i == n_iter - 1)
            scene_config.validate_gradients(opt, i)
            loss_values.append(float(loss))
            opt.step()
            scene_config.validate_params(opt, i)
            scene_config.update_scene(sdf_scene, i)
            params.update(opt)
            if rank == 0 and (i % 16 == 0 or i == n_iter - 1):
                print(f"[{i:4d}/{n_iter}] Loss: {loss_values[-1]:.4f}", flush=True)
    finally:
        if rank == 0:
            try:
                import matplotlib
                matplotlib.use('Agg')
                import matplotlib.pyplot as plt
                plt.figure()
                plt.plot(np.arange(len(loss_values)), loss_values)
                plt.xlabel('Iterations'); plt.ylabel('Objective function value')
                plt.savefig(join(output_dir, 'loss.png'))
            except Exception:
                np.save(join(output_dir, 'loss.npy'), np.array(loss_values))
            dump_metadata(config, scene_config, {'total_time': time.time() - t_start, 'loss_values': loss_values},
                          join(output_dir, 'metadata.json'))
    # exponential moving average of the parameters -> final checkpoint (shape_opt.py:126-129)
    if scene_config.param_averaging_beta is not None:
        scene_config.load_mean_parameters(opt)
        if rank == 0:
            scene_config.save_params(opt, output_dir, 'final')
        params.update(opt)
    integ.warp_field = None
    return loss_values
