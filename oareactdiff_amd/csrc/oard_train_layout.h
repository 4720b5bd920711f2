// oard_train_layout.h — transposed packs of the node-side matrices for the backward sweep (oard_train_stages.h).
// Included by oard_hip.hip after Packer / ParamIdx / BwdOff.
#pragma once

// ---- transposed packs of the node-side matrices (appended to the oard_pack_weights_bwd blob) -------------------------------------
struct NodeBwdLayerOff { size_t vpT, xv0T, xv2T, xp2T, xp0T, nm1T, nm0T, W1aT, W1bT; };
struct NodeBwdOff {
    NodeBwdLayerOff layer[OARD_MAX_LAYERS];
    size_t pe1T, un0T, v1pT, emboutT, un2T, s2vT, rl2T, rl0T, embT, nbembT;
    size_t total;          // floats, including `base`
};
static NodeBwdOff make_node_bwd_layout(const oard_config* c, size_t base) {
    const RDims d(c->hidden, c->num_radial);
    NodeBwdOff b;
    memset(&b, 0, sizeof(b));
    size_t cur = base;
    auto mat = [&](int MT, int KB) { size_t o = cur; cur = align_up(cur + (size_t)MT * KB * 256, 64); return o; };
    for (int l = 0; l < c->num_layers; ++l) {
        NodeBwdLayerOff& n = b.layer[l];
        n.vpT = mat(d.HT, 2 * d.HT); n.xv0T = mat(2 * d.HT, d.HT); n.xv2T = mat(d.HT, 3 * d.HT);
        n.xp2T = mat(d.HT, 3 * d.HT); n.xp0T = mat(d.HT, d.HT); n.nm1T = mat(d.HT, d.HT); n.nm0T = mat(2 * d.HT, d.HT);
        n.W1aT = mat(d.HT, d.HT); n.W1bT = mat(d.HT, d.HT);
    }
    b.pe1T = mat(d.PB, d.HT); b.un0T = mat(2 * d.HT, d.HT); b.v1pT = mat(d.HT, d.HT); b.emboutT = mat(d.HT, 1); b.un2T = mat(d.HT, 1);
    b.s2vT = mat(d.HT, d.HT); b.rl2T = mat(d.HT, d.HT); b.rl0T = mat(d.RB, d.HT); b.embT = mat(1, d.HT); b.nbembT = mat(1, d.HT);
    b.total = cur;
    return b;
}
static void pack_node_bwd(const oard_config* c, Packer& pk, const NodeBwdOff& nb) {
    const ParamIdx pi(c);
    const RDims d(c->hidden, c->num_radial);
    const int H = d.H, W = d.W, R = d.R, C = c->in_hidden;
    // transpose = 1: destination rows run over the SOURCE columns (the layer's input features), the K index over the source rows
    for (int l = 0; l < c->num_layers; ++l) {
        const NodeBwdLayerOff& n = nb.layer[l];
        const int g = pi.gcl0 + 14 * l, m = pi.msg0 + 9 * l, u = pi.upd0 + 9 * l;
        pk.matrix(u + 0, H, 0, H, d.HP, 1, H, d.HP, 2, d.HT, 2 * d.HT, n.vpT, 0, 256, 0, 1);          // vec_proj [2H][H]
        pk.matrix(u + 1, 2 * H, 0, H, d.HP, 2, H, d.HP, 1, 2 * d.HT, d.HT, n.xv0T, 0, 256, 0, 1);     // xvec_proj.0 [H][2H]
        pk.matrix(u + 2, H, 0, H, d.HP, 1, H, d.HP, 3, d.HT, 3 * d.HT, n.xv2T, 0, 256, 0, 1);          // xvec_proj.2 [3H][H]
        pk.matrix(m + 5, H, 0, H, d.HP, 1, H, d.HP, 3, d.HT, 3 * d.HT, n.xp2T, 0, 256, 0, 1);          // x_proj.2 [3H][H]
        pk.matrix(m + 4, H, 0, H, d.HP, 1, H, d.HP, 1, d.HT, d.HT, n.xp0T, 0, 256, 0, 1);              // x_proj.0 [H][H]
        pk.matrix(g + 6, H, 0, H, d.HP, 1, H, d.HP, 1, d.HT, d.HT, n.nm1T, 0, 256, 0, 1);              // node_mlp.1 [H][H]
        pk.matrix(g + 4, 2 * H, 0, H, d.HP, 2, H, d.HP, 1, 2 * d.HT, d.HT, n.nm0T, 0, 256, 0, 1);     // node_mlp.0 [H][2H]
        pk.matrix(g + 0, 2 * H + W, 0, H, d.HP, 1, H, d.HP, 1, d.HT, d.HT, n.W1aT, 0, 256, 0, 1);      // edge_mlp.0[:, 0:H]
        pk.matrix(g + 0, 2 * H + W, H, H, d.HP, 1, H, d.HP, 1, d.HT, d.HT, n.W1bT, 0, 256, 0, 1);      // edge_mlp.0[:, H:2H]
    }
    const int o = pi.out0;
    pk.matrix(pi.pe1_w, d.H2, 0, d.H2, d.PP, 1, H, d.HP, 1, d.PB, d.HT, nb.pe1T, 0, 256, 0, 1);        // pos_expansion.mlp.1 [H][H/2]
    pk.matrix(o + 2, 2 * H, 0, H, d.HP, 2, H, d.HP, 1, 2 * d.HT, d.HT, nb.un0T, 0, 256, 0, 1);         // update_net.0 [H][2H]
    pk.matrix(o + 0, H, 0, H, d.HP, 1, H, d.HP, 1, d.HT, d.HT, nb.v1pT, 0, 256, 0, 1);                 // vec1_proj [H][H]
    pk.matrix(pi.embout_w, H, 0, H, d.HP, 1, C, 16, 1, d.HT, 1, nb.emboutT, 0, 256, 0, 1);            // embedding_out [C][H]
    pk.matrix(o + 4, H, 0, H, d.HP, 1, 2, 16, 1, d.HT, 1, nb.un2T, 0, 256, 0, 1);                      // update_net.2 [2][H]
    pk.matrix(pi.s2v_w, H, 0, H, d.HP, 1, H, d.HP, 1, d.HT, d.HT, nb.s2vT, 0, 256, 0, 1);              // s2v.lin1.0 [H][H]
    pk.matrix(pi.rl2_w, H, 0, H, d.HP, 1, H, d.HP, 1, d.HT, d.HT, nb.rl2T, 0, 256, 0, 1);              // radial_lin.2 [H][H]
    pk.matrix(pi.rl0_w, R, 0, R, d.RP, 1, H, d.HP, 1, d.RB, d.HT, nb.rl0T, 0, 256, 0, 1);              // radial_lin.0 [H][R]
    pk.matrix(pi.emb_w, C, 0, C, 16, 1, H, d.HP, 1, 1, d.HT, nb.embT, 0, 256, 0, 1);                    // embedding [H][C]
    pk.matrix(pi.nbemb_w, C, 0, C, 16, 1, H, d.HP, 1, 1, d.HT, nb.nbembT, 0, 256, 0, 1);                // neighbor_emb.embedding [H][C]
}

