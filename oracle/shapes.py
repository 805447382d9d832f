"""State-dict key -> shape maps of the reference models, derived from config alone.

TEST INFRASTRUCTURE ONLY.  Restates the module constructors (qwen2_navit.py:381-398,687-705,943-959,
1095-1111; siglip_navit.py:145-165,262-269,330-343; bagel.py:61-86; modeling_utils.py:74-143;
autoencoder.py:38-272) so weights can be synthesised without the reference tree.  Checked against
the real reference ``state_dict()`` in tests/test_reference_crosscheck.py (build container only) and
against the product's modules in tests/test_state_dict.py.
"""


def bagel_shapes(cfg):
    llm, vit, bg = cfg["llm"], cfg["vit"], cfg["bagel"]
    H, I, V = llm["hidden_size"], llm["intermediate_size"], llm["vocab_size"]
    nh, nkv = llm["num_attention_heads"], llm["num_key_value_heads"]
    hd = H // nh
    s = {}
    lm = "language_model."
    s[lm + "model.embed_tokens.weight"] = (V, H)
    kind = llm.get("layer_module", "Qwen2MoTDecoderLayer")
    # Decoder_layer_dict (qwen2_navit.py:936-940): MoT duplicates attention + norms + MLP per modality (:381-398,687-705),
    # MoE only the MLP (:834-846), the dense layer nothing (:603-615)
    attn_sufs = ("", "_moe_gen") if kind == "Qwen2MoTDecoderLayer" else ("",)
    mlp_sufs = ("", "_moe_gen") if kind in ("Qwen2MoTDecoderLayer", "Qwen2MoEDecoderLayer") else ("",)
    for i in range(llm["num_hidden_layers"]):
        p = f"{lm}model.layers.{i}."
        for suf in attn_sufs:
            s[p + f"self_attn.q_proj{suf}.weight"] = (nh * hd, H)
            s[p + f"self_attn.q_proj{suf}.bias"] = (nh * hd,)
            s[p + f"self_attn.k_proj{suf}.weight"] = (nkv * hd, H)
            s[p + f"self_attn.k_proj{suf}.bias"] = (nkv * hd,)
            s[p + f"self_attn.v_proj{suf}.weight"] = (nkv * hd, H)
            s[p + f"self_attn.v_proj{suf}.bias"] = (nkv * hd,)
            s[p + f"self_attn.o_proj{suf}.weight"] = (H, nh * hd)
            s[p + f"self_attn.q_norm{suf}.weight"] = (hd,)
            s[p + f"self_attn.k_norm{suf}.weight"] = (hd,)
            s[p + f"input_layernorm{suf}.weight"] = (H,)
            s[p + f"post_attention_layernorm{suf}.weight"] = (H,)
        for suf in mlp_sufs:
            s[p + f"mlp{suf}.gate_proj.weight"] = (I, H)
            s[p + f"mlp{suf}.up_proj.weight"] = (I, H)
            s[p + f"mlp{suf}.down_proj.weight"] = (H, I)
    s[lm + "model.norm.weight"] = (H,)
    if "Mo" in kind:                      # Qwen2Model.use_moe (:948,957-958)
        s[lm + "model.norm_moe_gen.weight"] = (H,)
    s[lm + "lm_head.weight"] = (V, H)
    # Bagel glue
    pdim = bg["latent_patch_size"] ** 2 * cfg["vae"]["z_channels"]
    s["time_embedder.mlp.0.weight"] = (H, 256)
    s["time_embedder.mlp.0.bias"] = (H,)
    s["time_embedder.mlp.2.weight"] = (H, H)
    s["time_embedder.mlp.2.bias"] = (H,)
    s["vae2llm.weight"] = (H, pdim)
    s["vae2llm.bias"] = (H,)
    s["llm2vae.weight"] = (pdim, H)
    s["llm2vae.bias"] = (pdim,)
    s["latent_pos_embed.pos_embed"] = (bg["max_latent_size"] ** 2, H)
    s["vit_pos_embed.pos_embed"] = (bg["vit_max_num_patch_per_side"] ** 2, H)
    D, VI = vit["hidden_size"], vit["intermediate_size"]
    s["connector.fc1.weight"] = (H, D)
    s["connector.fc1.bias"] = (H,)
    s["connector.fc2.weight"] = (H, H)
    s["connector.fc2.bias"] = (H,)
    vp = "vit_model.vision_model."
    s[vp + "embeddings.patch_embedding.weight"] = (D, vit["num_channels"] * vit["patch_size"] ** 2)
    s[vp + "embeddings.patch_embedding.bias"] = (D,)
    if not vit.get("rope", False):
        s[vp + "embeddings.position_embedding.weight"] = ((vit["image_size"] // vit["patch_size"]) ** 2, D)
    else:   # RotaryEmbedding2D buffers (siglip_navit.py:102-127,337-340): persistent, hence state-dict entries
        side = vit["image_size"] // vit["patch_size"]
        for n in ("cos_h", "sin_h", "cos_w", "sin_w"):
            s[vp + "rope." + n] = (side * side, D // vit["num_attention_heads"] // 2)
    for i in range(vit["num_hidden_layers"]):
        p = f"{vp}encoder.layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            s[p + f"self_attn.{n}.weight"] = (D, D)
            s[p + f"self_attn.{n}.bias"] = (D,)
        for n in ("layer_norm1", "layer_norm2"):
            s[p + n + ".weight"] = (D,)
            s[p + n + ".bias"] = (D,)
        s[p + "mlp.fc1.weight"] = (VI, D)
        s[p + "mlp.fc1.bias"] = (VI,)
        s[p + "mlp.fc2.weight"] = (D, VI)
        s[p + "mlp.fc2.bias"] = (D,)
    s[vp + "post_layernorm.weight"] = (D,)
    s[vp + "post_layernorm.bias"] = (D,)
    return s


def vae_shapes(v):
    s = {}

    def conv(p, cin, cout, k):
        s[p + ".weight"] = (cout, cin, k, k)
        s[p + ".bias"] = (cout,)

    def gn(p, c):
        s[p + ".weight"] = (c,)
        s[p + ".bias"] = (c,)

    def res(p, cin, cout):
        gn(p + ".norm1", cin)
        conv(p + ".conv1", cin, cout, 3)
        gn(p + ".norm2", cout)
        conv(p + ".conv2", cout, cout, 3)
        if cin != cout:
            conv(p + ".nin_shortcut", cin, cout, 1)

    def attn(p, c):
        gn(p + ".norm", c)
        for n in ("q", "k", "v", "proj_out"):
            conv(p + "." + n, c, c, 1)

    ch, mult, nb, z = v["ch"], v["ch_mult"], v["num_res_blocks"], v["z_channels"]
    nres = len(mult)
    # encoder
    conv("encoder.conv_in", v["in_channels"], ch, 3)
    in_mult = (1,) + tuple(mult)
    bi = ch
    for lvl in range(nres):
        bi, bo = ch * in_mult[lvl], ch * mult[lvl]
        for b in range(nb):
            res(f"encoder.down.{lvl}.block.{b}", bi, bo)
            bi = bo
        if lvl != nres - 1:
            conv(f"encoder.down.{lvl}.downsample.conv", bi, bi, 3)
    res("encoder.mid.block_1", bi, bi)
    attn("encoder.mid.attn_1", bi)
    res("encoder.mid.block_2", bi, bi)
    gn("encoder.norm_out", bi)
    conv("encoder.conv_out", bi, 2 * z, 3)
    # decoder
    bi = ch * mult[nres - 1]
    conv("decoder.conv_in", z, bi, 3)
    res("decoder.mid.block_1", bi, bi)
    attn("decoder.mid.attn_1", bi)
    res("decoder.mid.block_2", bi, bi)
    for lvl in reversed(range(nres)):
        bo = ch * mult[lvl]
        for b in range(nb + 1):
            res(f"decoder.up.{lvl}.block.{b}", bi, bo)
            bi = bo
        if lvl != 0:
            conv(f"decoder.up.{lvl}.upsample.conv", bi, bi, 3)
    gn("decoder.norm_out", bi)
    conv("decoder.conv_out", bi, v["out_ch"], 3)
    return s
