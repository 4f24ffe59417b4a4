(* GPHIP.wl -- thin Wolfram-Language host package for the MI355X GP path.  Load AFTER the reference package
   (BayesianInference/Kernel/BayesianInference.wl:11-19): it reuses the reference's inferenceObject, $MachineLogZero,
   dataNormalForm, defineInferenceProblem, evidenceSampling and swaps (1) the "LogLikelihoodFunction" closure (seam at
   BayesianGaussianProcess.wl:249, 293-294) and (2) the prediction down-value for HIP-backed objects (:343-376), keeping every
   key of "GaussianProcessData"/"ModelFunctions" in the SHAPE the reference defines (:257-262, 308, 314-321).
   All numerics AND all host logic that does not need a Wolfram kernel (kernel-name grammar, CForm text -> C function body,
   starting pool of a tabulated prior) are in libgphip (include/gphip.h) behind the LibraryLink shim (csrc/librarylink_shim.cpp)
   and run under test there (tests/test_host_logic.py, tests/test_gpu_wl_shim.py).  This file cannot be executed in the build
   containers: tests/test_wl_package.py checks every LibraryFunctionLoad against the shim's exports and argument counts, and
   every call made here is mirrored 1:1 by bayesianinference_amd/gaussian_process.py, which the parity tests exercise. *)

BeginPackage["GPHIP`", {"BayesianUtilities`", "BayesianStatistics`", "BayesianGaussianProcess`"}]

defineGaussianProcessHIP::usage = "defineGaussianProcessHIP[X -> Y, kernel, nugget, meanFunction, variables, prior, opts] has the argument list of defineGaussianProcess (BayesianGaussianProcess.wl:228-234) and builds the same inferenceObject with the log-likelihood evaluated on the GPU. kernel: a named kernel \"SE\", \"SEARD\", \"Matern52\", \"Matern52ARD\", \"Matern32\", \"Matern32ARD\", \"RQ\", \"RQARD\", a composed form \"term + term\", \"term * term\", optionally followed by \" + Const\" (e.g. \"SE + Const\"), or None (null kernel); ANY OTHER kernel expression falls through to the reference's own defineGaussianProcess. nugget: \"Constant\" (Function[sn^2]) or any expression / function of the point in the parameter symbols (evaluated on the host per theta, the values go to the GPU). meanFunction: None, \"Constant\" or any expression / function of the point. variables in the library's order: {term 1: l.., (alpha), sf}, {term 2 ..}, {c}, {sn}, {mu}. The short form defineGaussianProcessHIP[X -> Y, kernelName, variables, prior, opts] takes the constant nugget and \"ConstantMean\" -> False | True. Options: \"Precision\" -> \"Double\" | \"Single\", \"Devices\" -> Automatic | {0, 1, ..}, \"LibraryOptions\" -> {\"panel\" -> 4, ..}.";
nestedSamplingHIP::usage = "nestedSamplingHIP[obj, opts] runs the native batched nested-sampling driver of the library (lock-step walkers: one batched likelihood call per Metropolis step) on a HIP-backed GP object whose prior is a product of UniformDistribution's (or \"PriorKinds\" -> {0 | 1 ..}, 1 = log-uniform over the parameter's range) or ANY product of univariate distributions (its factors' log densities travel as tables, the starting pool is drawn from the prior here) and returns the object joined with the result, in the shape nestedSampling returns (the reference's own evidenceSampling post-processes the samples). Takes the options of nestedSampling plus \"Walkers\" -> 32 and \"Seed\" -> 0. Joint (non-separable) priors fall through to nestedSampling[obj], which drives the same GPU closure one theta at a time.";
defineGaussianProcessHIP::nonnative = "Kernel `1` is not one of the named or composed kernels of the library: give the nugget and the mean function as expressions in the parameter symbols (not \"Constant\"); the kernel is then compiled for the device from its CForm, or runs on the reference's own path if it cannot be printed as C.";
hipKernelFunction::usage = "hipKernelFunction[kernelName, d] gives theta |-> Function[{p, q}, ..], the exact WL form of the named kernel (what one would hand to the reference's defineGaussianProcess for the same model).";
$GPHIPLibrary::usage = "Path of the LibraryLink shim (libgphip_wl).";

Begin["`Private`"]

$GPHIPLibrary = FindLibrary["libgphip_wl"];
termNames = {"SE", "SEARD", "Matern52", "Matern52ARD", None, "Matern32", "Matern32ARD", "RQ", "RQARD"};      (* by library id, gphip.h *)
(* kernel spec -> {term1, op (0 none, 1 sum, 2 product), term2 | None, offset (0 | 1), kernelId, n1, n2, snIndex (1-based)} or
   $Failed (not a kernel of the library).  The grammar -- term, "term + term", "term * term", optionally "+ Const" -- is parsed
   by the library (gphip_kernel_parse), the same code the Python host and the tests use. *)
parseKernel[None, d_] := parseKernel["None", d];
parseKernel[name_String, d_Integer] := With[{s = gpKernelSpec[name, d]},
	If[ !VectorQ[s, IntegerQ] || First[s] < 0, $Failed,
		{termNames[[s[[2]] + 1]], s[[3]], If[s[[4]] < 0, None, termNames[[s[[4]] + 1]]], s[[5]], s[[1]], s[[6]], s[[7]], s[[8]] + 1}]];
parseKernel[__] := $Failed;
kernelCode[spec_] := spec[[5]];
nuggetIndex[spec_, d_] := spec[[8]];
nLengthScales[name_, d_] := If[StringEndsQ[name, "ARD"], d, 1];

(* ---- LibraryLink bindings (names and argument counts are checked against the shim by tests/test_wl_package.py) ---- *)
load[name_, args_, ret_] := LibraryFunctionLoad[$GPHIPLibrary, name, args, ret];
m2 = {Real, 2, "Constant"}; v1 = {Real, 1, "Constant"}; any = {Real, _, "Constant"}; iv = {Integer, 1, "Constant"};     (* {} = "not given" fits `any` *)
gpCreate := gpCreate = load["gphip_wl_create", {m2, v1, Integer, Integer, Integer, iv}, Integer];
gpCreateCustom := gpCreateCustom = load["gphip_wl_create_custom", {m2, v1, "UTF8String", Integer, Integer, Integer, iv}, Integer];
gpKernelSpec := gpKernelSpec = load["gphip_wl_kernel_spec", {"UTF8String", Integer}, {Integer, 1}];
gpSetOpt := gpSetOpt = load["gphip_wl_set_option", {Integer, "UTF8String", Real}, Integer];
gpLogLik := gpLogLik = load["gphip_wl_loglik", {Integer, v1}, {Real, 1}];                          (* {value, info} *)
gpLogLikB := gpLogLikB = load["gphip_wl_loglik_batch", {Integer, m2}, {Real, 2}];                  (* {{value, info}..} *)
gpGrad := gpGrad = load["gphip_wl_loglik_grad", {Integer, v1}, {Real, 1}];                         (* {value, info, grad..} *)
gpFit := gpFit = load["gphip_wl_fit", {Integer, v1}, Integer];                                     (* info *)
gpSolve := gpSolve = load["gphip_wl_solve", {Integer, any}, {Real, _}];                            (* vector (N) or matrix (N x m) *)
gpLogDet := gpLogDet = load["gphip_wl_logdet", {Integer}, Real];
gpPredict := gpPredict = load["gphip_wl_predict", {Integer, m2}, {Real, 2}];                       (* {means, variances} *)
gpPredictS := gpPredictS = load["gphip_wl_predict_samples", {Integer, m2, m2}, {Real, 3}];         (* {means, variances}, each S x M *)
gpCov := gpCov = load["gphip_wl_covariance", {Integer, v1}, {Real, 2}];
gpCross := gpCross = load["gphip_wl_cross_covariance", {Integer, v1, m2}, {Real, 2}];              (* (N+1) x M: k on top, kappa last row *)
gpDestroy := gpDestroy = load["gphip_wl_destroy", {Integer}, Integer];
gpDevices := gpDevices = load["gphip_wl_device_count", {}, Integer];
(* point-dependent nugget[x] / meanFunction[x]: VALUES per theta and point; {} = the constant form *)
gpLogLikBPW := gpLogLikBPW = load["gphip_wl_loglik_batch_pw", {Integer, m2, any, any}, {Real, 2}];
gpFitPW := gpFitPW = load["gphip_wl_fit_pw", {Integer, v1, any, any}, Integer];
gpPredictSPW := gpPredictSPW = load["gphip_wl_predict_samples_pw", {Integer, m2, any, any, m2, any, any}, {Real, 3}];

(* value -> machine real; info != 0 or a LibraryFunctionError -> $MachineLogZero, exactly what
   Catch[..., "MatInv"] yields in the reference closure (BayesianGaussianProcess.wl:298-304). *)
toLogLik[{val_Real, info_Real}] := If[info == 0., Clip[val, {-Abs[$MachineLogZero], Abs[$MachineLogZero]}], $MachineLogZero];
toLogLik[_] := $MachineLogZero;

(* ---- the named kernels as the reference would see them: theta |-> pure function ---- *)
profile["SE"][r2_] := Exp[-r2/2];
profile["Matern52"][r2_] := With[{s = Sqrt[5 r2]}, (1 + s + s^2/3) Exp[-s]];
profile["Matern32"][r2_] := With[{s = Sqrt[3 r2]}, (1 + s) Exp[-s]];
profile["RQ"][r2_, a_] := (1 + r2/(2 a))^(-a);
(* theta slice of one term: {l.. (1 or d), (alpha), sf};  r2 = Total[((p - q)/l)^2] *)
hipKernelFunction[name_String, d_] := With[{nl = nLengthScales[name, d], base = StringDelete[name, "ARD"]},
	Function[theta, With[{ls = theta[[;; nl]], more = theta[[nl + 1 ;; -2]], sf = theta[[-1]]},
		Function[{p, q}, sf^2 profile[base][Total[((p - q)/ls)^2], Sequence @@ more]]]]];
hipKernelFunction[None, d_] := Function[theta, Function[0]];      (* nullKernelPattern, BayesianGaussianProcess.wl:25 *)
(* composed forms: each term reads its own slice of theta; c sits behind the terms *)
hipKernelFunction[{None, __}, d_] := hipKernelFunction[None, d];
hipKernelFunction[{t1_String, 0, _, 0, _, n1_, __}, d_] := Function[theta, hipKernelFunction[t1, d][theta[[;; n1]]]];
hipKernelFunction[{t1_String, op_, t2_, offset_, _, n1_, n2_, _}, d_] :=
	Function[theta,
		With[{
			k1 = hipKernelFunction[t1, d][theta[[;; n1]]],
			k2 = If[op === 0, None, hipKernelFunction[t2, d][theta[[n1 + 1 ;; n1 + n2]]]],
			c = If[offset === 1, theta[[n1 + n2 + 1]], 0]
		},
			Switch[op, 0, Function[{p, q}, c + k1[p, q]], 1, Function[{p, q}, c + k1[p, q] + k2[p, q]], 2, Function[{p, q}, c + k1[p, q] k2[p, q]]]
		]
	];
hipNuggetFunction[spec_, d_] := With[{i = nuggetIndex[spec, d]},
	Function[theta, With[{sn = theta[[i]]}, Function[sn^2]]]];
hipMeanFunction[spec_, d_, False] := Function[theta, Function[0]];
hipMeanFunction[spec_, d_, True] := With[{i = nuggetIndex[spec, d] + 1},
	Function[theta, With[{mu = theta[[i]]}, Function[mu]]]];

(* the handle keeps ONE factor resident: refit only when theta changed; every other call reuses the workspace, so it first
   forgets the fit (touch).  fit = gpFit[h, #]& or the point-dependent form that also hands over nugget / mean values. *)
$fitted = <||>;
touch[h_] := ($fitted[h] = None);
ensureFit[h_, theta_, fit_] := If[ Lookup[$fitted, h, None] === theta, 0,
	With[{info = fit[theta]}, $fitted[h] = If[info === 0, theta, None]; info]];

(* ---- an arbitrary kernel as C text.  The kernel expression (in the parameter symbols) is applied to two points of stand-in
   symbols gphipXc<k> / gphipYc<k>, the parameter symbols become gphipPc<k> in the order of `variables`; the result must be free
   of anything CForm cannot print as elementary arithmetic.  ToString[CForm[..]] goes to the library AS IT IS: the shim turns the
   stand-ins into X(k) / Y(k) / P(k) and compiles the text (gphip_wl_create_custom).  The library's theta for such a handle is
   {p.., sn}: the reference's nugget arrives as values per point, so the sn slot is a dummy 1. appended by customLift. ---- *)
customKernelSpec[kerf_, vars_List, d_Integer] := Module[{xs, ys, ps, expr},
	xs = Table[Symbol["GPHIP`Private`gphipXc" <> ToString[k]], {k, 0, d - 1}];
	ys = Table[Symbol["GPHIP`Private`gphipYc" <> ToString[k]], {k, 0, d - 1}];
	ps = Table[Symbol["GPHIP`Private`gphipPc" <> ToString[k]], {k, 0, Length[vars] - 1}];
	expr = Quiet @ Check[kerf[xs, ys], $Failed];
	If[ expr === $Failed || !FreeQ[expr, _Function | _Slot | _Piecewise | _If | _Which | _List | _Dot], Return[$Failed]];
	expr = N[expr /. Thread[vars -> ps]];
	(* anything left that is neither a stand-in nor a System` function cannot be printed as C *)
	If[ Cases[expr, sym_Symbol /; !MemberQ[Join[xs, ys, ps], sym] && Context[sym] =!= "System`", {0, Infinity}, Heads -> True] =!= {},
		Return[$Failed]];
	{"Custom", ToString[CForm[expr]], kerf, Length[vars], vars}
];
customSpecQ[spec_] := MatchQ[spec, {"Custom", _String, _, _Integer, _List}];
customLift[spec_][theta_] := If[ customSpecQ[spec],
	If[MatrixQ[theta], ArrayFlatten[{{theta, ConstantArray[1., {Length[theta], 1}]}}], Join[theta, {1.}]],
	theta
];
hipKernelFunction[{"Custom", _, kerf_, _, vars_}, d_] := Function[theta, kerf /. Thread[vars -> theta]];
hipNuggetFunction[{"Custom", __}, d_] := Function[theta, Function[1.]];       (* the dummy sn^2 the library puts on the diagonal *)

Options[defineGaussianProcessHIP] = {"ConstantMean" -> False, "Precision" -> "Double", "Devices" -> Automatic, "LibraryOptions" -> {}};

constantQ[f_] := MatchQ[f, "Constant" | Automatic];
zeroMeanQ[f_] := MatchQ[f, None | 0 | 0. | Function[0] | (0 &)];

(* The reference's own argument list (BayesianGaussianProcess.wl:228-234).  A kernel that is neither a named / composed form
   nor printable as C falls through to the reference's defineGaussianProcess with the caller's arguments untouched. *)
defineGaussianProcessHIP[dataIn_List?(MatrixQ[#, NumericQ]&) -> dataOut_List?(MatrixQ[#, NumericQ]&), kerf_, nugf_, meanf_,
	variables : {{_Symbol, _, _}..}, variablePrior_, rest___Rule
] /; Dimensions[dataOut][[2]] === 1 && Length[dataIn] === Length[dataOut] := Module[{
	spec = parseKernel[kerf, Dimensions[dataIn][[2]]],
	own = "ConstantMean" | "Precision" | "Devices" | "LibraryOptions"
},
	If[ spec === $Failed,
		(* not a named kernel: every argument has the reference's meaning (no "Constant" shorthands) *)
		If[ StringQ[nugf] || nugf === Automatic || StringQ[meanf],
			Message[defineGaussianProcessHIP::nonnative, kerf]; Return[inferenceObject[$Failed]]];
		(* ANY pure function of two points in the parameter symbols (BGP:29-33): printed with CForm and compiled by the library
		   at run time into its own kernel build.  Only an expression that is not elementary arithmetic of the coordinates,
		   or a text that does not compile, takes the reference's interpreted path.  (hipGaussianProcess frees its handle
		   itself when the object cannot be completed.) *)
		With[{custom = customKernelSpec[kerf, variables[[All, 1]], Dimensions[dataIn][[2]]]},
			If[ custom =!= $Failed,
				With[{obj = hipGaussianProcess[dataIn -> dataOut, custom, nugf, meanf, variables, variablePrior, rest]},
					If[ !MatchQ[obj, inferenceObject[$Failed]], Return[obj]]]]
		];
		Return @ defineGaussianProcess[dataIn -> dataOut, kerf, nugf, meanf, variables, variablePrior,
			Sequence @@ FilterRules[{rest}, Except[own]]]
	];
	hipGaussianProcess[dataIn -> dataOut, spec, nugf, meanf, variables, variablePrior, rest]
];
(* short form: named kernel, constant nugget, "ConstantMean" option *)
defineGaussianProcessHIP[data : (_List -> _List), kernelName : (_String | None), variables : {{_Symbol, _, _}..}, variablePrior_, rest___Rule] :=
	defineGaussianProcessHIP[data, kernelName, "Constant", If[TrueQ[Lookup[{rest}, "ConstantMean", False]], "Constant", None], variables, variablePrior, rest];
defineGaussianProcessHIP[___] := inferenceObject[$Failed];

hipGaussianProcess[dataIn_ -> dataOut_, spec_, nugf_, meanf_, variables_, variablePrior_, rest___Rule] := Module[{
	h, loglik, invCov, fit, values,
	d = Dimensions[dataIn][[2]],
	vars = variables[[All, 1]],
	inputData = Developer`ToPackedArray[N @ dataIn],
	constMean = constantQ[meanf],
	nugPW = !constantQ[nugf],                               (* nugget[points[[i]]] evaluated on the host (BGP:37) *)
	meanPW = !constantQ[meanf] && !zeroMeanQ[meanf],         (* meanFunction /@ inputData (BGP:300) *)
	nugget, mean, pw,
	dtype = If[Lookup[{rest}, "Precision", "Double"] === "Single", 32, 64],
	(* sub-kernels of parallelNestedSampling pick their own GPU (BayesianStatistics.wl:1349); several ordinals = ONE
	   multi-device handle (the library shards a large factorisation over them) *)
	devices = Replace[Lookup[{rest}, "Devices", Automatic], {Automatic :> {Mod[$KernelID, Max[gpDevices[], 1]]}, i_Integer :> {i}}],
	own = "ConstantMean" | "Precision" | "Devices" | "LibraryOptions"
},
	h = If[ customSpecQ[spec],
		Quiet @ gpCreateCustom[inputData, N @ Flatten[dataOut], spec[[2]], spec[[4]], 0, dtype, devices],      (* (a text that does not compile: LibraryFunctionError) *)
		gpCreate[inputData, N @ Flatten[dataOut], kernelCode[spec], Boole[constMean], dtype, devices]
	];
	If[ !IntegerQ[h] || h < 0, Return[inferenceObject[$Failed]]];
	KeyValueMap[gpSetOpt[h, #1, N[#2]]&, Association @ Lookup[{rest}, "LibraryOptions", {}]];
	(* theta |-> function of the point, exactly as the reference builds them (expressionToFunction, BGP:257-262) *)
	nugget = If[nugPW, expressionToFunction[nugf, vars -> paramVector], hipNuggetFunction[spec, d]];
	mean = Which[meanPW, expressionToFunction[meanf, vars -> paramVector], True, hipMeanFunction[spec, d, constMean]];
	pw = nugPW || meanPW;
	(* values of a point function for every theta of a batch: B x Length[pts], or {} for the constant form *)
	values[f_, on_, thetas_, pts_] := If[on, Developer`ToPackedArray @ N @ Map[Function[th, f[th] /@ pts], thetas], {}];
	loglik = Function[theta,
		touch[h];
		Which[
			pw, With[{ths = If[MatrixQ[theta], N @ theta, {N @ theta}]},
				With[{res = toLogLik /@ gpLogLikBPW[h, customLift[spec][ths], values[mean, meanPW, ths, inputData], values[nugget, nugPW, ths, inputData]]},
					If[MatrixQ[theta], res, First[res]]]],
			MatrixQ[theta], toLogLik /@ gpLogLikB[h, N @ theta],
			True, toLogLik @ gpLogLik[h, N @ theta]
		]
	];
	fit = If[ pw,
		Function[th, gpFitPW[h, customLift[spec][th], Flatten @ values[mean, meanPW, {th}, inputData], Flatten @ values[nugget, nugPW, {th}, inputData]]],
		Function[th, gpFit[h, th]]
	];
	(* matrixInverseAndDet[covarianceFunction[theta]] (BayesianGaussianProcess.wl:130-141, 308): a solver for a vector or a
	   matrix (:194, :410, :416) and the log-determinant; singular K Throws the sentinel with tag "MatInv" exactly like :133 *)
	invCov = Function[theta, With[{th = N @ theta},
		If[ ensureFit[h, th, fit] =!= 0, Throw[$MachineLogZero, "MatInv"]];
		<|"Inverse" -> Function[b, If[ensureFit[h, th, fit] =!= 0, Throw[$MachineLogZero, "MatInv"]]; gpSolve[h, N @ b]], "LogDet" -> gpLogDet[h]|>]];
	freeOnFailure[h] @ defineInferenceProblem[            (* same keys as BayesianGaussianProcess.wl:310-325 *)
		"Data" -> dataNormalForm[dataIn -> dataOut],
		"PriorDistribution" -> variablePrior,
		"Parameters" -> variables,
		"GaussianProcessData" -> <|
			"ModelFunctions" -> <|
				"KernelFunction" -> hipKernelFunction[spec, d],
				"NuggetFunction" -> nugget,
				"MeanFunction" -> mean,
				(* the library builds K with the constant sn^2 on the diagonal; a point-dependent nugget replaces it here *)
				"CovarianceFunction" -> Function[theta, touch[h];
					With[{K = gpCov[h, customLift[spec][N @ theta]]},
						If[nugPW, K + DiagonalMatrix[(nugget[theta] /@ inputData) - hipNuggetFunction[spec, d][theta][]], K]]],
				"InverseCovarianceFunction" -> invCov
			|>,
			"KernelName" -> spec,
			"PointwiseFunctions" -> {meanPW, nugPW},
			"ThetaLift" -> customLift[spec],                       (* identity for the named kernels *)
			"HIPHandle" -> h
		|>,
		Sequence @@ FilterRules[{rest}, Except[own]],
		"LogLikelihoodGradientFunction" -> If[pw, Missing["PointDependentFunctions"], Function[theta, touch[h]; gpGrad[h, N @ theta]]],
		"LogLikelihoodFunction" -> loglik
	]
];

(* an object that could not be completed (BS:276-298 smoke test, a bad prior, ..) must not leak its GPU handle *)
freeOnFailure[h_][obj_] := (If[MatchQ[obj, inferenceObject[$Failed]] || !MatchQ[obj, _inferenceObject], gpDestroy[h]]; obj);

(* ---- prediction for HIP-backed objects: same return shape as BayesianGaussianProcess.wl:343-376 ---- *)
hipObjectQ = Function[AssociationQ[#] && KeyExistsQ[#, "Samples"] && KeyExistsQ[Lookup[#, "GaussianProcessData", <||>], "HIPHandle"]];
hipPredict[result_, pts_List] := Module[{
	h = result["GaussianProcessData", "HIPHandle"],
	points = dataNormalForm[pts],
	train = result["Data"][[1]],
	weights = Values @ result[["Samples", All, "CrudePosteriorWeight"]],
	thetas = N @ Values @ result[["Samples", All, "Point"]],
	mf = result["GaussianProcessData", "ModelFunctions"],
	pwFlags = Lookup[result["GaussianProcessData"], "PointwiseFunctions", {False, False}],
	vals, perSample
},
	If[ Dimensions[points][[2]] =!= Dimensions[train][[2]], Return[$Failed]];       (* test points of the wrong width *)
	vals[f_, on_, at_] := If[on, Developer`ToPackedArray @ N @ Map[Function[th, f[th] /@ at], thetas], {}];
	(* one batched call: every posterior sample is factored and solved in its own workspace slot (singular K: NaN rows);
	   point-dependent nugget / mean functions are evaluated per sample at the training AND the test points (BGP:113, 408) *)
	perSample = With[{mv = If[ Or @@ pwFlags,
			gpPredictSPW[h, Lookup[result["GaussianProcessData"], "ThetaLift", Identity][thetas], vals[mf["MeanFunction"], pwFlags[[1]], train], vals[mf["NuggetFunction"], pwFlags[[2]], train],
				N @ points, vals[mf["MeanFunction"], pwFlags[[1]], points], vals[mf["NuggetFunction"], pwFlags[[2]], points]],
			gpPredictS[h, thetas, N @ points]
		]},
		MapThread[Function[{mus, vars}, MapThread[NormalDistribution, {mus, Sqrt[vars]}]], {mv[[1]], mv[[2]]}]];
	touch[h];                                                (* the batched pass reused the handle's workspace *)
	AssociationThread[points, MixtureDistribution[weights, #]& /@ Transpose[perSample]]
];

(* The reference's own definition (BayesianGaussianProcess.wl:343-346) matches a HIP object just as well (the LHS differs only
   inside a PatternTest, which WL cannot order by specificity) and was defined first: the HIP rule is PREPENDED. *)
Unprotect[predictFromGaussianProcess];
DownValues[predictFromGaussianProcess] = Prepend[DownValues[predictFromGaussianProcess],
	HoldPattern[predictFromGaussianProcess[inferenceObject[result_?hipObjectQ], pts_List]] :> hipPredict[result, pts]];

(* predictiveDistribution (BayesianStatistics.wl:1373-1387) needs a "GeneratingDistribution", which a GP object does
   not carry; for HIP-backed GP objects it forwards to the prediction above.  The "MaximumLikelihood" / "MAP"
   forms (:1389-1416) reduce "Samples" to one element and re-enter here. *)
bestSample[result_, f_] := Append[result, "Samples" -> TakeLargestBy[result["Samples"], f, 1]];
Unprotect[predictiveDistribution];
DownValues[predictiveDistribution] = Join[{
	HoldPattern[predictiveDistribution[inferenceObject[result_?hipObjectQ], pts_List]] :> hipPredict[result, pts],
	HoldPattern[predictiveDistribution[inferenceObject[result_?hipObjectQ], pts_List, "MaximumLikelihood"]] :> hipPredict[bestSample[result, #LogLikelihood &], pts],
	HoldPattern[predictiveDistribution[inferenceObject[result_?hipObjectQ], pts_List, "MAP"]] :> hipPredict[bestSample[result, #LogLikelihood + #LogPriorPDF &], pts]
}, DownValues[predictiveDistribution]];

(* the native batched sampler for these objects (nestedSamplingHIP, row f1 of the scope table) is a package of its own *)
Get[FileNameJoin[{DirectoryName[$InputFileName], "GPHIPSampler.wl"}]];

End[]
EndPackage[]
