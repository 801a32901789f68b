"""Output-side boundary: PFM byte layout, camera text layout, model switch."""
import numpy as np
import pytest

from deep3d_aerial_amd import predict


def test_pfm_byte_layout(tmp_path):
    img = np.arange(12, dtype=np.float32).reshape(3, 4) / 3
    path = str(tmp_path / "a.pfm")
    predict.save_pfm(path, img)
    raw = open(path, "rb").read()
    # data_io.py:196-223: 'Pf', 'W H', '-1.000000' (little endian), rows bottom-up
    assert raw.startswith(b"Pf\n4 3\n-1.000000\n")
    body = np.frombuffer(raw[len(b"Pf\n4 3\n-1.000000\n"):], "<f4").reshape(3, 4)
    assert np.array_equal(body, img[::-1])
    back, scale = predict.load_pfm(path)
    assert scale == 1.0 and np.array_equal(back, img)
    with pytest.raises(Exception, match="float32"):
        predict.save_pfm(path, img.astype(np.float64))


def test_cam_text_layout(tmp_path):
    cam = np.zeros((2, 4, 4), np.float32)
    cam[0] = np.eye(4)
    cam[1, :3, :3] = [[1000, 0, 320], [0, 1000, 240], [0, 0, 1]]
    cam[1, 3] = [400.0, 1.25, 384, 880.0]
    path = str(tmp_path / "c.txt")
    predict.write_red_cam(path, cam, ["640", "480", "7", "img.png"], "/x/img.png")
    lines = open(path).read().split("\n")
    assert lines[0] == "extrinsic: XrightYdown, [Rcw|tcw]"
    assert lines[1].split() == ["1.0", "0.0", "0.0", "0.0"]
    assert lines[5] == "" and lines[6] == "intrinsic"
    assert lines[7].split() == ["1000.0", "0.0", "320.0"]
    assert lines[11].split() == ["400.0", "1.25", "384.0", "880.0"]
    assert lines[13] == "640 480 7 img.png /x/img.png"


def test_model_switch():
    assert type(predict.build_model("casmvsnet", 64)).__name__ == "Infer_CascadeMVSNet"
    assert type(predict.build_model("adamvs", 64)).__name__ == "Infer_AdaMVSNet"
    with pytest.raises(Exception, match="Not implemented yet"):
        predict.build_model("nonsense", 64)


def test_synthetic_block_item_layout():
    ds = predict.SyntheticBlock(3, 3, 64, 96, 64)
    s = ds[1]
    assert s["imgs"].shape == (3, 3, 64, 96) and s["depth_values"].shape == (2,)
    p1, p3 = s["proj_matrices"]["stage1"], s["proj_matrices"]["stage3"]
    assert np.allclose(p1[:, :2], p3[:, :2] / 4) and np.array_equal(p1[:, 2:], p3[:, 2:])


# ---- row N4: launch of the inference step and fusion set-up (mvs/mvs_dl.py:27-65, run.py:98-108,176) ----------------
def test_mvs_inference_arguments_and_errors(tmp_path):
    from deep3d_aerial_amd import mvs_dl

    inf = mvs_dl.MVS_Inference(768, 384, view_num=3, num_depth=64, model_type="CasMVSNet", pretrain_weight="w.ckpt")
    argv = inf.argv("/data/block", str(tmp_path / "mvs"))
    assert argv[:3] == ["--data_folder=/data/block", "--output_folder=%s" % (tmp_path / "mvs"), "--model=casmvsnet"]
    assert "--view_num=3" in argv and "--numdepth=64" in argv and "--max_w=768" in argv and "--max_h=384" in argv
    assert "--min_interval=0.1" in argv and "--display=False" in argv and "--loadckpt=w.ckpt" in argv
    a = predict.parse_args(argv)  # the harness accepts exactly what the launcher formats
    assert (a.model, a.view_num, a.numdepth, a.max_w, a.max_h, a.loadckpt) == ("casmvsnet", 3, 64, 768, 384, "w.ckpt")
    with pytest.raises(Exception, match="Not implemented yet"):
        mvs_dl.MVS_Inference(768, 384, model_type="rednet").run("/data/block", str(tmp_path / "x" / "mvs"))


def test_multi_rank_launch_propagates_failure(tmp_path):
    """N > 1: one worker per rank through torch.distributed.run; a failing worker raises here instead of being
    dropped like os.system's status (mvs_dl.py:65).  The workers fail by design: no GPU / no dataset reader."""
    from deep3d_aerial_amd import mvs_dl

    inf = mvs_dl.MVS_Inference(96, 64, view_num=3, num_depth=32, model_type="casmvsnet", n_gpus=2,
                               extra_args=["--random_weights"])
    with pytest.raises(RuntimeError, match="exit status"):
        inf.run("/nonexistent", str(tmp_path / "mvs"))


def test_missing_checkpoint_is_an_error_not_random_weights(tmp_path):
    """mvs_dl.py:52-58 / predict.py:105: the reference fails when no checkpoint exists; products of an untrained network
    must not be written silently."""
    from deep3d_aerial_amd import mvs_dl

    inf = mvs_dl.MVS_Inference(96, 64, view_num=3, num_depth=32, model_type="casmvsnet")
    with pytest.raises(FileNotFoundError, match="no checkpoint"):
        inf.run(str(tmp_path), str(tmp_path / "mvs"))
    with pytest.raises(FileNotFoundError, match="--loadckpt is required"):
        predict.main(["--data_folder", str(tmp_path), "--output_folder", str(tmp_path / "o")])
    with pytest.raises(ValueError, match="--data_folder is required"):
        predict.main(["--output_folder", str(tmp_path / "o"), "--random_weights"])


def test_harness_accepts_every_reference_flag():
    """predict.py:31-56: all 21 flags parse, with the reference's defaults."""
    a = predict.parse_args(["--output_folder", "o"])
    assert (a.model, a.dataset, a.view_num, a.numdepth, a.max_w, a.max_h) == ("adamvs", "cas_normal_eval", 5, 192, 3584, 4096)
    assert (a.fext, a.normalize, a.resize_scale, a.sample_scale, a.interval_scale, a.batch_size) == (".jpg", "mean", 1.0, 1, 1, 1)
    assert (a.share_cr, a.ndepths, a.depth_inter_r, a.cr_base_chs, a.min_interval) == (False, "48,32,8", "4,2,1", "8,8,8", 0.1)
    b = predict.parse_args(["--model", "casmvsnet", "--dataset", "cas_normal_eval", "--data_folder", "d", "--output_folder", "o",
                            "--loadckpt", "w", "--view_num", "3", "--numdepth", "64", "--max_w", "96", "--max_h", "64",
                            "--min_interval", "0.2", "--fext", ".png", "--normalize", "standard", "--resize_scale", "0.5",
                            "--sample_scale", "1", "--interval_scale", "2", "--batch_size", "1", "--display", "False",
                            "--share_cr", "--ndepths", "8,8,4", "--depth_inter_r", "2,1,1", "--cr_base_chs", "8,8,8"])
    assert b.share_cr and b.normalize == "standard" and b.resize_scale == 0.5 and b.interval_scale == 2
    assert predict._truthy(b.display) is False and predict._truthy("True") and predict._truthy(True)


def test_fusion_settings_honour_every_threshold():
    import yaml
    from deep3d_aerial_amd import mvs_dl

    cfg = yaml.safe_load("""
FUSION:
  run_depth_fusion: true
  fusion_num: 7
  geo_consist_num: 3
  photomatric_threshold: 0.35
  position_threshold: 0.5
  depth_threshold: 0.02
  normal_threshold: 30.0
  pc_format: "ply"
""")
    s = mvs_dl.fusion_settings(cfg)
    assert (s["fusion_num"], s["min_geo_consist_num"], s["photometric_threshold"]) == (7, 3, 0.35)
    assert (s["position_threshold"], s["depth_threshold"], s["normal_threshold"]) == (0.5, 0.02, 30.0)
