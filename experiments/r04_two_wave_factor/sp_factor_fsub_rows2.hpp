// archived experiment (round 4), not compiled: see README.md
// The same factorisation on TWO wavefronts of a workgroup (k_lm_run2 / k_lm_spec: the second wavefront of a capsule is idle while the system is solved). The trailing update of
// column k is split: the items whose target lies in column k+1 -- the ones the next pivot and panel wait for; the host lists them first (b == 0: targets in the column of the
// parent, which is k+1 wherever the elimination tree is a chain) -- stay on the first wavefront, all the others (targets in columns >= k+2) go to the second one, which runs one
// column behind: while it subtracts the outer products of column k, the first wavefront is already on the pivot and panel of column k+1 (blocks of column k+1 only: disjoint).
// One workgroup barrier per column orders the two: [pivot, panel of k | lazy updates of k-1] barrier [urgent updates of k | lazy updates of k]. Every target block still
// receives its contributions in the order of the columns they come from, each from one lane: the factor is bit-identical to sp_factor_fsub_rows'. `flag`: one int of LDS.
__device__ __forceinline__ bool sp_factor_fsub_rows2(const SparseSys &S, int *flag) {
	const int lane = threadIdx.x & 63, nb = S.nb; const bool second = threadIdx.x >= 64;
	const int grp = (lane * 171) >> 9, sub = lane - 3 * grp; // lane / 3, lane % 3 for lane < 64
	const bool worker = lane < 63;
	int cb = S.col_off[0], ce = nb > 0 ? S.col_off[1] : cb, ib = 0;
	int ra = (!second && worker && cb + grp < ce) ? S.row[cb + grp] : 0; // block-row of this lane's panel block in the coming column
	for (int k = 0; k < nb; k++) {
		const int cn = ce - cb, nitems = cn * (cn + 1) / 2;
		const int ce_n = (k + 2 <= nb) ? S.col_off[k + 2] : ce;
		const int nu = (cn > 0 && S.row[cb] == k + 1) ? cn : 0; // the first cn items of a column target the column of its parent
		int ra_n = 0; bool ok = true;
		if (!second) {
			double *D = S.diag + 9 * k;
			const double a00 = D[0], a10 = D[3], a11 = D[4], a20 = D[6], a21 = D[7], a22 = D[8];
			const double b0 = S.rhs[3 * k], b1 = S.rhs[3 * k + 1], b2 = S.rhs[3 * k + 2];
			const bool pl = worker && grp < cn;
			double *Arow = S.off + 9 * (cb + grp) + 3 * sub; double *rr = S.rhs + 3 * ra + sub;
			double A0 = 0, A1 = 0, A2 = 0, rv = 0;
			if (pl) { A0 = Arow[0]; A1 = Arow[1]; A2 = Arow[2]; rv = *rr; }
			ra_n = (worker && ce + grp < ce_n) ? S.row[ce + grp] : 0; // index load for the next column
			Chol3 c;
			ok = chol3v(a00, a10, a11, a20, a21, a22, c);
			if (ok) {
				const double y0 = b0 * c.r0, y1 = (b1 - c.l10 * y0) * c.r1, y2 = (b2 - c.l20 * y0 - c.l21 * y1) * c.r2;
				if (pl) {
					const double x0 = A0 * c.r0, x1 = (A1 - x0 * c.l10) * c.r1, x2 = (A2 - x0 * c.l20 - x1 * c.l21) * c.r2;
					Arow[0] = x0; Arow[1] = x1; Arow[2] = x2;
					*rr = rv - (x0 * y0 + x1 * y1 + x2 * y2);
				}
				if (worker) for (int p = grp + 21; p < cn; p += 21) { // columns with more than 21 blocks
					double *Ax = S.off + 9 * (cb + p) + 3 * sub; double *rx = S.rhs + 3 * S.row[cb + p] + sub;
					const double x0 = Ax[0] * c.r0, x1 = (Ax[1] - x0 * c.l10) * c.r1, x2 = (Ax[2] - x0 * c.l20 - x1 * c.l21) * c.r2;
					Ax[0] = x0; Ax[1] = x1; Ax[2] = x2; *rx -= x0 * y0 + x1 * y1 + x2 * y2;
				}
				if (lane == SRBA_WG - 1) { // L_kk, reciprocal diagonal in the unused upper part, y_k
					D[0] = c.l00; D[3] = c.l10; D[4] = c.l11; D[6] = c.l20; D[7] = c.l21; D[8] = c.l22; D[1] = c.r0; D[2] = c.r1; D[5] = c.r2;
					S.rhs[3 * k] = y0; S.rhs[3 * k + 1] = y1; S.rhs[3 * k + 2] = y2;
				}
			}
			if (lane == 0) *flag = ok ? 1 : 0;
		}
		__syncthreads();
		if (second) ok = *flag != 0;
		if (!ok) return false;
		const int t0 = second ? nu : 0, t1 = second ? nitems : nu;
		if (worker) for (int t = t0 + grp; t < t1; t += 21) { // trailing update, row `sub` of target -= L_ak L_bk^t
			const unsigned w = (unsigned)S.item[ib + t];
			const double *La = S.off + 9 * (cb + ((w >> 9) & 511)) + 3 * sub, *Lb = S.off + 9 * (cb + (w & 511)); double *T = S.diag + 9 * (w >> 18) + 3 * sub;
			const double la0 = La[0], la1 = La[1], la2 = La[2];
			double lb[9];
#pragma unroll
			for (int q = 0; q < 9; q++) lb[q] = Lb[q];
			const double t0_ = T[0], t1_ = T[1], t2_ = T[2];
			T[0] = t0_ - (la0 * lb[0] + la1 * lb[1] + la2 * lb[2]);
			T[1] = t1_ - (la0 * lb[3] + la1 * lb[4] + la2 * lb[5]);
			T[2] = t2_ - (la0 * lb[6] + la1 * lb[7] + la2 * lb[8]);
		}
		if (!second) solver_sync();
		cb = ce; ce = ce_n; ib += nitems; ra = ra_n;
	}
	return true;
}
